"""Where a GEMM tile's time goes: an INSTRUMENTED build of csrc/gemm.hip whose workgroups
stamp the 100-MHz wall clock at four points of every output tile (K loop begins, K loop ends,
epilogue begins, epilogue ends; the shader-clock counter as well around the K loop) into a
device array; this script builds it into tools/probes/bin/libmi355dit_timing.so (never the
product library), runs one launch per shape and prints per-phase medians, the clock inside the
K loop and how far apart the workgroups of a round are.  Extra compiler flags after `build`
select ablations of the ring kernel (timing only, the results are wrong): -DABL_RING_NOMFMA
(refill + barrier + fragment reads alone), -DABL_RING_NODMA (no refill), -DABL_EPI_NOSTORE (the
epilogues without their global stores).

    python tools/probes/gemm_phase_timing.py build      # here, no GPU
    python tools/probes/gemm_phase_timing.py run         # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "sglang-diffusion_amd", "csrc")
OUT = os.environ.get("TIMING_LIB", os.path.join(ROOT, "tools", "probes", "bin", "libmi355dit_timing.so"))
MAXT = 64          # tiles per workgroup recorded
NWG = 512


def build():
    s = open(os.path.join(CSRC, "gemm.hip")).read()

    def rep(old, new, count=1):
        nonlocal s
        assert s.count(old) == count, (old, s.count(old))
        s = s.replace(old, new)

    rep("namespace {\n\nstruct GemmParams {",
        "__device__ unsigned long long g_stamp[%d * %d * 4];\n"
        "__device__ unsigned long long g_cyc[%d * %d * 2];\n"
        "extern \"C\" int mi355_dbg_cycles(unsigned long long* host) {\n"
        "  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cyc), sizeof(g_cyc));\n}\n"
        "extern \"C\" int mi355_dbg_stamps(unsigned long long* host) {\n"
        "  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(g_stamp));\n}\n"
        "namespace {\n\nstruct GemmParams {" % (NWG, MAXT, NWG, MAXT))
    rep("  while (true) {\n    AccTile<MF16, TN, TM> acct;",
        "  int tile_no = 0;\n  while (true) {\n"
        "    if (tid == 0 && tile_no < %d) { g_stamp[(blockIdx.x * %d + tile_no) * 4 + 0] = wall_clock64();"
        " g_cyc[(blockIdx.x * %d + tile_no) * 2 + 0] = clock64(); }\n"
        "    AccTile<MF16, TN, TM> acct;" % (MAXT, MAXT, MAXT))
    rep("    const int lastb = (g - 1) & 1;\n",
        "    if (tid == 0 && tile_no < %d) { g_stamp[(blockIdx.x * %d + tile_no) * 4 + 1] = wall_clock64();"
        " g_cyc[(blockIdx.x * %d + tile_no) * 2 + 1] = clock64(); }\n"
        "    const int lastb = (g - 1) & 1;\n" % (MAXT, MAXT, MAXT))
    rep("    __syncthreads();  // tables visible\n    if constexpr (MF16)",
        "    __syncthreads();  // tables visible\n"
        "    if (tid == 0 && tile_no < %d) g_stamp[(blockIdx.x * %d + tile_no) * 4 + 2] = wall_clock64();\n"
        "    if constexpr (MF16)" % (MAXT, MAXT))
    rep("    if (!more) break;\n    vb = nvb;\n",
        "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n"
        "    if (tid == 0 && tile_no < %d) g_stamp[(blockIdx.x * %d + tile_no) * 4 + 3] = wall_clock64();\n"
        "    ++tile_no;\n    if (!more) break;\n    vb = nvb;\n" % (MAXT, MAXT))
    # the ring kernel: stamps, and -DABL_RING_NOMFMA (its K loop without the MFMAs)
    rep("  while (true) {\n    f32x4 acc[TN * 2][TM * 2];",
        "  int tile_no = 0;\n  while (true) {\n"
        "    if (tid == 0 && tile_no < %d) { g_stamp[(blockIdx.x * %d + tile_no) * 4 + 0] = wall_clock64();"
        " g_cyc[(blockIdx.x * %d + tile_no) * 2 + 0] = clock64(); }\n"
        "    f32x4 acc[TN * 2][TM * 2];" % (MAXT, MAXT, MAXT), count=s.count("  while (true) {\n    f32x4 acc[TN * 2][TM * 2];"))
    rep("    const int s_scr = next_slot(sl, NSLOT - 2), s_tab = next_slot(sl, NSLOT - 1);\n",
        "    if (tid == 0 && tile_no < %d) { g_stamp[(blockIdx.x * %d + tile_no) * 4 + 1] = wall_clock64();"
        " g_cyc[(blockIdx.x * %d + tile_no) * 2 + 1] = clock64(); }\n"
        "    const int s_scr = next_slot(sl, NSLOT - 2), s_tab = next_slot(sl, NSLOT - 1);\n" % (MAXT, MAXT, MAXT))
    rep("    gemm_epilogue16<1, 4, 2, 2, 4>(p, acc, smem + s_scr * SLOT",
        "    if (tid == 0 && tile_no < %d) g_stamp[(blockIdx.x * %d + tile_no) * 4 + 2] = wall_clock64();\n"
        "    gemm_epilogue16<1, 4, 2, 2, 4>(p, acc, smem + s_scr * SLOT" % (MAXT, MAXT))
    rep("    if (!more) {\n      asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");  // the unused tail",
        "    if (tid == 0 && tile_no < %d) g_stamp[(blockIdx.x * %d + tile_no) * 4 + 3] = wall_clock64();\n"
        "    ++tile_no;\n"
        "    if (!more) {\n      asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");  // the unused tail" % (MAXT, MAXT),
        count=s.count("    if (!more) {\n      asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");  // the unused tail"))
    rep("""            acc[ig + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                af[i], bfr[j], acc[ig + i][j], 0, 0, 0, unit_scale, 0, unit_scale);
          __builtin_amdgcn_sched_barrier(0);""", """#ifdef ABL_RING_NOMFMA
            asm volatile("" ::"v"(af[i]), "v"(bfr[j]));
#else
            acc[ig + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                af[i], bfr[j], acc[ig + i][j], 0, 0, 0, unit_scale, 0, unit_scale);
#endif
          __builtin_amdgcn_sched_barrier(0);""")
    # -DABL_RING_NODMA: the ring without its refill
    rep("""  auto piece_a = [&](int slot, int i) {
    __builtin_amdgcn_global_load_lds(""", """  auto piece_a = [&](int slot, int i) {
#ifdef ABL_RING_NODMA
    if (slot >= 0) return;
#endif
    __builtin_amdgcn_global_load_lds(""")
    rep("""  auto piece_w = [&](int slot, int i) {
    __builtin_amdgcn_global_load_lds(""", """  auto piece_w = [&](int slot, int i) {
#ifdef ABL_RING_NODMA
    if (slot >= 0) return;
#endif
    __builtin_amdgcn_global_load_lds(""")
    # -DABL_EPI_NOSTORE: the epilogues without their global stores (what the transposition and
    # the arithmetic cost alone; a tile's 128 KiB leave every CU at the same moment otherwise)
    st = "__builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(p.out + orow + n8));"
    rep(st, "\n#ifdef ABL_EPI_NOSTORE\n          if (p.M < 0)\n#endif\n          " + st, count=s.count(st))
    os.makedirs("/tmp/gemm_timing", exist_ok=True)
    src = "/tmp/gemm_timing/gemm_timing.hip"
    open(src, "w").write(s)
    flags = sys.argv[2:]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950",
                           "-Wno-unused-result", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), *flags,
                           "-c", src, "-o", "/tmp/gemm_timing/gemm_timing.o"])
    objs = [os.path.join(CSRC, "build", f) for f in ("quant.o", "attention.o", "elementwise.o", "api.o")]
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT,
                           "/tmp/gemm_timing/gemm_timing.o", *objs])
    print("built", OUT)


def ktiles_of(K, fp8):
    return K * (1 if fp8 else 2) // 128


def run():
    os.environ["MI355_DIT_LIB"] = OUT
    sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
    import numpy as np
    import torch
    from sglang_diffusion_amd import _C

    lib = ctypes.CDLL(OUT)
    dev = "cuda"
    shapes = [(True, 16384, 9216, 3072), (True, 18432, 12288, 3072), (True, 18432, 3072, 15360),
              (False, 16384, 12288, 3072), (False, 16384, 3072, 12288)]
    if os.environ.get("TIMING_FP8_ONLY"):
        shapes = shapes[:1] + shapes[2:3]
    if os.environ.get("TIMING_BF16_ONLY"):
        shapes = shapes[3:]
    for fp8, M, N, K in shapes:
        if fp8:
            a = (torch.randn(M, K, device=dev) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
            w = (torch.randn(N, K, device=dev) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
            sa = torch.rand(M, device=dev) * 0.01
            sb = torch.rand(N, device=dev) * 0.01
        else:
            a = torch.randn(M, K, device=dev).bfloat16()
            w = torch.randn(N, K, device=dev).bfloat16() * 0.02
            sa = sb = None
        bias = torch.randn(N, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        # TIMING_MODE: which epilogue the launches use (plain | act | f16 | gate)
        mode = os.environ.get("TIMING_MODE", "plain")
        extra = {}
        if mode == "act":
            extra = dict(act=1)
        elif mode == "f16":
            extra = dict(f16_cols=(N // 3 * 2, N))
        elif mode == "gate":
            gate = torch.randn(4, N, device=dev).bfloat16()
            out.normal_()
            extra = dict(residual=out, gate=gate, ldg=gate.stride(0), gate_rows=M // 4)
        for _ in range(int(os.environ.get("TIMING_WARM", "40"))):
            _C.linear_ex(a, w, out, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, sa=sa, sb=sb, bias=bias, fp8=fp8,
                         **extra)
        if mode == "f16":
            _C.f16_range_status(reset=True)
        torch.cuda.synchronize()
        host = np.zeros(NWG * MAXT * 4, dtype=np.uint64)
        rc = lib.mi355_dbg_stamps(host.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)))
        assert rc == 0, rc
        st = host.reshape(NWG, MAXT, 4).astype(np.float64) * 0.01      # us (100 MHz)
        hc = np.zeros(NWG * MAXT * 2, dtype=np.uint64)
        assert lib.mi355_dbg_cycles(hc.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))) == 0
        cyc = hc.reshape(NWG, MAXT, 2).astype(np.float64)
        tiles = (M // 256) * (N // 256)
        nwg = min(tiles, 256)
        per = -(-tiles // nwg)
        full = tiles // nwg                                              # rounds every workgroup runs
        s = st[:nwg, :full]
        kloop = s[..., 1] - s[..., 0]
        seam = s[..., 2] - s[..., 1]
        epi = s[..., 3] - s[..., 2]
        t0 = s[..., 0].min()
        kcyc = cyc[:nwg, :full, 1] - cyc[:nwg, :full, 0]
        print(f"    shader clock inside the K loop: {np.median(kcyc / kloop):.0f} MHz "
              f"({np.median(kcyc) / ktiles_of(K, fp8):.0f} clocks per K tile)", flush=True)
        ktiles = K * (1 if fp8 else 2) // 128
        print(f"{'fp8' if fp8 else 'bf16'} M={M} N={N} K={K}: {tiles} tiles, {per} rounds; per tile (median us): "
              f"K loop {np.median(kloop):.2f} ({np.median(kloop) / ktiles * 1e3:.0f} ns per K tile), "
              f"seam {np.median(seam):.2f}, epilogue+drain {np.median(epi):.2f}; "
              f"launch {s[..., 3].max() - t0:.1f} us; start spread of round r (max-min over workgroups): "
              + ", ".join(f"{s[:, r, 0].max() - s[:, r, 0].min():.1f}" for r in range(min(full, 6))), flush=True)
        # first and later rounds separately: cold caches against warm
        print("    K loop by round: " + ", ".join(f"{np.median(kloop[:, r]):.1f}" for r in range(full))
              + "   epilogue by round: " + ", ".join(f"{np.median(epi[:, r]):.1f}" for r in range(full)), flush=True)


if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)()
