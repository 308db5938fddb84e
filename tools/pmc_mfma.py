"""Matrix-pipe occupancy per kernel from one rocprofv3 counter pass.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES \\
              SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 bench.py ...
    python tools/pmc_mfma.py DIR > profiles/rNN_pmc_mfma.json

GRBM_GUI_ACTIVE is summed over the 8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8.
SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (32 per v_mfma_f32_32x32x16_bf16, 64 per
32x32x64 f8f6f4), summed over the chip's 1024 SIMDs; SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES
count quad-cycles.  mfma_busy = MFMA_BUSY / (1024 * kernel cycles)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short  # noqa: E402

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"normalisation": "cycles = GRBM_GUI_ACTIVE / 8; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * "
                        "cycles); valu_active = 4 * SQ_ACTIVE_INST_VALU / (1024 * cycles); coexec likewise",
       "kernels": {}}
for k, c in sorted(acc.items()):
    mean = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = mean.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    d = {"dispatches": max(len(v) for v in c.values()), "counters_avg": mean}
    if cyc > 0:
        d["kernel_cycles"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
            d["mfma_busy"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        if "SQ_ACTIVE_INST_VALU" in mean:
            d["valu_active"] = 4.0 * mean["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc)
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in mean:
            d["valu_mfma_coexec"] = mean["SQ_VALU_MFMA_COEXEC_CYCLES"] / (1024.0 * cyc)
    out["kernels"][k] = d
json.dump(out, sys.stdout, indent=1)
