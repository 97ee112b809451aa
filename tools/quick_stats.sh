#!/bin/bash
# Run on the GPU box (gpurun): ONE rocprofv3 kernel-trace pass of bench.py with every launch on one stream (VXM_NO_OVERLAP=1), summarised per
# kernel into gpurun_out/TAG_kernel_stats_serial.csv -- the quick per-kernel view between two edits (tools/profile_bench.sh is the full set).
# usage: tools/quick_stats.sh TAG [bench args]
set -u
TAG=${1:-q}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
VXM_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-extra-configs "$@" > $OUT/bench.log 2>&1
python tools/rocprof_summary.py stats $OUT/stats_serial gpurun_out/${TAG}_kernel_stats_serial.csv
rm -rf $OUT/stats_serial
head -45 gpurun_out/${TAG}_kernel_stats_serial.csv
