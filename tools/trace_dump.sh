#!/bin/bash
# GPU box helper: rocprofv3 kernel trace of a short bench.py run (replayed steps), step anatomy + the dispatch list of the last step
# usage: tools/trace_dump.sh TAG [bench args]   -> gpurun_out/TAG_dispatch.txt, gpurun_out/TAG_overlap.txt
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-extra-configs "$@" > $OUT/bench.log 2>&1
rm -f gpurun_out/${TAG}_dispatch.txt
python tools/trace_overlap.py $OUT/t 5 6 --dump gpurun_out/${TAG}_dispatch.txt > gpurun_out/${TAG}_overlap.txt 2>&1
rm -rf $OUT/t
cat gpurun_out/${TAG}_overlap.txt | tail -20
