#!/bin/bash
# Alternate two builds of the library on the attention timing and on the whole step, one box.
# usage: bash tools/probes/ab_attn.sh [ALT_LIB]   ("base" = the shipped library, "alt" = ALT_LIB through MI355_DIT_LIB;
# default ALT_LIB: tools/probes/bin/libmi355dit_before.so)
ALT=${1:-/root/repo/tools/probes/bin/libmi355dit_before.so}
for i in 1 2; do
  echo "== alt"; MI355_DIT_LIB=$ALT timeout -k 10 120 python -u tools/attn_time.py 2>&1 | grep attention || exit 1
  echo "== base"; timeout -k 10 120 python -u tools/attn_time.py 2>&1 | grep attention || exit 1
done
for i in 1 2; do
  echo "== alt step"; MI355_DIT_LIB=$ALT timeout -k 10 200 python -u bench.py --no-c4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['breakdown']['attention']['avg_launch_us'])" || exit 1
  echo "== base step"; timeout -k 10 200 python -u bench.py --no-c4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['breakdown']['attention']['avg_launch_us'])" || exit 1
done
