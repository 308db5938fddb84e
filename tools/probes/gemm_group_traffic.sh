#!/bin/bash
# L2-memory-side traffic of one 16384 x 9216 x 3072 fp8 GEMM for several tile-walk group sizes
# (separate counter passes; MI355_GEMM_GROUP is read by the library per call and inherited from this shell)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for g in 8 4 2 16; do
  export MI355_GEMM_GROUP=$g
  O=gpurun_out/grp_$g
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 tools/gemm_one.py fp8 > /dev/null 2> $O.f.err || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 tools/gemm_one.py fp8 > /dev/null 2> $O.w.err || exit 1
  echo "group $g"
  python3 tools/pmc_traffic.py $O/f $O/w | python3 -c "
import json,sys
d=json.load(sys.stdin)['kernels']
for k,v in d.items():
    if k.startswith('gemm_kernel'): print('  ', k, round(v['traffic_bytes_per_launch']/1e6,1), 'MB per launch (fetch KiB', round(v['FETCH_SIZE_KB_avg']), 'write KiB', round(v['WRITE_SIZE_KB_avg']), ')')
"
  rm -rf $O
done
