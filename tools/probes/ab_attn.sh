#!/bin/bash
# alternate old/new library on attention timing and step time
OLD=/root/repo/tools/probes/bin/libmi355dit_before.so
for i in 1 2; do
  echo "== old"; MI355_DIT_LIB=$OLD timeout -k 10 120 python -u tools/attn_time.py 2>&1 | grep attention || exit 1
  echo "== new"; timeout -k 10 120 python -u tools/attn_time.py 2>&1 | grep attention || exit 1
done
for i in 1 2; do
  echo "== old step"; MI355_DIT_LIB=$OLD timeout -k 10 200 python -u bench.py --no-c4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['breakdown']['attention']['avg_launch_us'])" || exit 1
  echo "== new step"; timeout -k 10 200 python -u bench.py --no-c4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['breakdown']['attention']['avg_launch_us'])" || exit 1
done
