#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats + HBM traffic counters (separate --pmc passes, kernel-trace only)
# of the SAME bench.py command; summaries land in gpurun_out/ and are copied to profiles/ by hand (tracked).
# usage: tools/profile_bench.sh TAG [bench args]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--steps 4 --warmup 2 --no-cpu-baseline $*"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py $ARGS > $OUT/bench_stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py $ARGS > $OUT/bench_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py $ARGS > $OUT/bench_write.log 2>&1
# calibration of FETCH_SIZE (= RDREQ x 64 B on gfx950): memory-side read requests by size class
timeout 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/rdreq -- python bench.py $ARGS > $OUT/bench_rdreq.log 2>&1
python tools/rocprof_summary.py stats $OUT/stats gpurun_out/${TAG}_rocprof_kernel_stats.csv
python tools/rocprof_summary.py pmc $OUT/fetch $OUT/write $OUT/rdreq gpurun_out/${TAG}_hbm_counters.json
grep -h '^{' $OUT/bench_stats.log > gpurun_out/${TAG}_bench_under_rocprof.json
# raw traces are large: keep only the summaries
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/rdreq
