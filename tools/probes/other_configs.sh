#!/bin/bash
# the other configurations of the bench, one JSON line each (one box)
set -o pipefail
OUT=${1:-gpurun_out/other_configs.jsonl}
: > $OUT
for args in "--batch 1" "--batch 2" "--batch 8" "--schnell" "--mode fp8-block" "--mode bf16"; do
  timeout -k 10 300 python3 -u bench.py --no-c4 --no-cpu-baseline $args 2>/dev/null | tail -1 >> $OUT || exit 1
  echo "done: $args"
done
python3 - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(round(d["ms_per_step"], 1), round(d["value"], 4), d["config"]["workload"][-60:])
PY
