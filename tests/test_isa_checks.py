"""Build-time properties of the compiled kernels, read from the code objects inside
libmi355dit.so (no GPU needed).

The attention kernel and the pipelined bf16 GEMM read their LDS fragments through inline asm with
hand-counted `s_waitcnt lgkmcnt(N)`: the compiler considers an asm output ready right behind the
statement, so a spill or re-materialisation of such a register between the read and the manual
wait would consume data LDS has not delivered yet.  A kernel of that kind must therefore not
spill vector registers at all.  The ring GEMM counts `vmcnt` by hand inside its K loops (a
scratch access there would both shift the count and drain the fill pipe): its K loops must be
free of scratch traffic; a register parked across an epilogue is only slow."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "sglang-diffusion_amd", "sglang_diffusion_amd", "libmi355dit.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernel_metadata(tmp_path):
    """{kernel name: {vgpr_count, vgpr_spill_count, ...}} over every gfx950 code object of the
    library (one bundle per translation unit, back to back in .hip_fatbin)."""
    fat = tmp_path / "fat.bin"
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", LIB])
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = {}
    _kernel_metadata.code_objects = []
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        part, co = tmp_path / f"bundle{n}.bin", tmp_path / f"co{n}.o"
        part.write_bytes(blob[a:b])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}",
                               f"--output={co}"])
        _kernel_metadata.code_objects.append(str(co))
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", str(co)]).decode()
        # kernel records are YAML maps with sorted keys: .agpr_count opens one, .wavefront_size ends it
        name, cur = None, {}
        for line in notes.splitlines():
            m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", line)
            if not m:
                continue
            key, val = m.group(1), m.group(2).strip()
            if key in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
                cur[key] = int(val)
            elif key == "name" and line.startswith("    .name"):      # the kernel's, not an argument's
                name = val.strip("'\"")
            elif key == "wavefront_size":
                if name:
                    out[name] = cur
                name, cur = None, {}
    return out


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    if not os.path.exists(f"{LLVM}/clang-offload-bundler"):
        pytest.skip("no LLVM tools in this image")
    md = _kernel_metadata(tmp_path_factory.mktemp("co"))
    assert len(md) > 40, sorted(md)[:5]
    return md


def test_kernels_with_hand_counted_lds_waits_do_not_spill(kernels):
    checked = 0
    for name, k in kernels.items():
        asm_reads = ("attn_fwd_kernel" in name                      # K / V fragment reads
                     or ("gemm_kernel" in name and name.endswith("Lb1ELb1EEEvNS_10GemmParamsE")))  # PIPE
        if asm_reads:
            checked += 1
            assert k.get("vgpr_spill_count", 0) == 0, (name, k)
    assert checked >= 4, checked     # attention instances, the pipelined GEMMs


def test_ring_gemm_k_loops_hold_no_scratch_traffic(kernels):
    """Disassemble the ring kernels: MFMAs come in runs (one K tile's worth, unrolled); no
    scratch_load / scratch_store may sit inside a run or between two runs that are close enough
    to be one loop body (a tile's epilogue is thousands of instructions away)."""
    found = 0
    for co in _kernel_metadata.code_objects:
        text = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co]).decode()
        if "gemm_ring_kernel" not in text:
            continue
        for fn in re.split(r"\n(?=[0-9a-f]+ <)", text):
            head = fn.split("\n", 1)[0]
            if "gemm_ring_kernel" not in head:
                continue
            found += 1
            ops = [ln.split()[0] for ln in fn.splitlines()[1:] if ln.startswith(" ") or ln.startswith("\t")]
            mfma = [i for i, o in enumerate(ops) if o.startswith("v_mfma")]
            assert len(mfma) >= 96, (head, len(mfma))
            spans, s0, prev = [], mfma[0], mfma[0]
            for i in mfma[1:]:
                if i - prev > 600:          # the next K loop (or the same one, unrolled again)
                    spans.append((s0, prev))
                    s0 = i
                prev = i
            spans.append((s0, prev))
            scratch = [i for i, o in enumerate(ops) if o.startswith("scratch_")]
            inside = [(i, sp) for i in scratch for sp in spans if sp[0] - 150 <= i <= sp[1] + 60]
            assert not inside, (head, inside[:4], spans)
    assert found == 2, found         # with and without the half-tile tail


def test_register_budgets_match_the_launch_shapes(kernels):
    """Two waves per SIMD (256 registers) for the 8-wave GEMMs and the attention kernel."""
    for name, k in kernels.items():
        if "gemm_ring_kernel" in name or "attn_fwd_kernel" in name or "gemm_kernel" in name:
            assert k["vgpr_count"] <= 256, (name, k)
