#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats + HBM traffic counters + MFMA-pipe counters (separate --pmc passes,
# kernel-trace only) of the SAME bench.py command; summaries land in gpurun_out/ and are copied to profiles/ by hand (tracked).
# usage: tools/profile_bench.sh TAG [bench args]      (VXM_PROFILE_SUFFIX=_bf16 with --config dense_bf16: bench.py looks for *_hbm_counters_bf16.json)
set -u
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
STEPS=4; WARM=2
# bench.py runs warm-up + the timed pass + one eager step that re-populates the allocator after the capture + the per-kernel pass
# (min(steps, 10) more steps): all of them are dispatches the profiler sees (a replayed hipGraph dispatches the same kernels)
export VXM_PROFILED_STEPS=$((STEPS + WARM + 1 + (STEPS < 10 ? STEPS : 10)))
SUF=${VXM_PROFILE_SUFFIX:-}
ARGS="--steps $STEPS --warmup $WARM --no-cpu-baseline --no-gpu-baseline --no-extra-configs $*"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py $ARGS > $OUT/bench_stats.log 2>&1
# the same, with every launch on ONE stream (VXM_NO_OVERLAP=1): the sum of kernel time is the step, small kernels are not inflated by sharing
# the chip with the second stream's weight-gradient launches
VXM_NO_OVERLAP=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- python bench.py $ARGS > $OUT/bench_stats_serial.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py $ARGS > $OUT/bench_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py $ARGS > $OUT/bench_write.log 2>&1
# MFMA utilisation of the conv kernels: SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipe is busy, summed over SIMDs),
# SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE (kernel time base), SQ_INSTS_MFMA, SQ_WAVE_CYCLES (quad-cycles)
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python bench.py $ARGS > $OUT/bench_mfma.log 2>&1
python tools/rocprof_summary.py stats $OUT/stats gpurun_out/${TAG}_rocprof_kernel_stats${SUF}.csv
python tools/rocprof_summary.py stats $OUT/stats_serial gpurun_out/${TAG}_rocprof_kernel_stats_serial${SUF}.csv
grep -h '^{' $OUT/bench_stats_serial.log > gpurun_out/${TAG}_bench_under_rocprof_serial${SUF}.json
python tools/rocprof_summary.py pmc $OUT/fetch $OUT/write gpurun_out/${TAG}_hbm_counters${SUF}.json
python tools/rocprof_summary.py pmc $OUT/mfma gpurun_out/${TAG}_mfma_counters${SUF}.json
grep -h '^{' $OUT/bench_stats.log > gpurun_out/${TAG}_bench_under_rocprof${SUF}.json
# raw traces are large: keep only the summaries
rm -rf $OUT/stats $OUT/stats_serial $OUT/fetch $OUT/write $OUT/mfma
