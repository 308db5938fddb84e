#!/bin/bash
# Final artefacts of a state of the kernels: bench line, kernel-trace statistics, HBM traffic counters.
# usage (on the GPU box, from the repo root): bash tools/probes/refresh_profiles.sh TAG
set -o pipefail
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 420 python3 -u bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "bench done" 
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o b4 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-c4 --no-overlap-text > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err || exit 1
echo "trace done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c4 --no-overlap-text > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c4 --no-overlap-text > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
echo "write done"
DB=$(find $OUT/prof -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB $OUT/kernel_stats.csv > $OUT/kernel_stats_top.txt
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
# the raw databases and counter CSVs are large: keep the summaries only
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/kernel_stats_top.txt
