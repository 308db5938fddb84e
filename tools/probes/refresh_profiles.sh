#!/bin/bash
# Final artefacts of a state of the kernels: bench line, kernel-trace statistics, HBM traffic counters,
# matrix-pipe / vector-pipe busy counters -- all from one call on one box.
# usage (on the GPU box, from the repo root): bash tools/probes/refresh_profiles.sh TAG
set -o pipefail
TAG=${1:-r03}
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
# matrix-pipe occupancy: 5 SQ counters (of 8 slots) + 1 GRBM, a pass of its own
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c4 --no-overlap-text > $OUT/pmc_mfma_run.json 2> $OUT/pmc_mfma.err || exit 1
echo "mfma done"
DB=$(find $OUT/prof -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB $OUT/kernel_stats.csv > $OUT/kernel_stats_top.txt
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
python3 tools/pmc_mfma.py $OUT/pmc_mfma > $OUT/pmc_mfma.json
# the raw databases and counter CSVs are large: keep the summaries only
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
cat $OUT/kernel_stats_top.txt
