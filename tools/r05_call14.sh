#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-gpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B $EXTRA > gpurun_out/r05n_$tag.json 2> gpurun_out/r05n_$tag.err; python - $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/r05n_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
    print("%-24s value %.2f  ms %.3f  host %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
EXTRA=""
run warm VXM_GRAPH=0
run graph VXM_GRAPH=1
run eager VXM_GRAPH=0
run graph2 VXM_GRAPH=1
run eager2 VXM_GRAPH=0
run graph3 VXM_GRAPH=1
EXTRA="--config dense_bf16"
run bf16_graph VXM_GRAPH=1
run bf16_eager VXM_GRAPH=0
run bf16_graph2 VXM_GRAPH=1
EXTRA="--batch-per-gpu 4 --steps 6 --warmup 4"
run b4_graph VXM_GRAPH=1
run b4_eager VXM_GRAPH=0
rm -rf gpurun_out/r05n_trace gpurun_out/r05n_dispatch_graph.txt
VXM_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05n_trace -- python bench.py --steps 6 --warmup 4 --no-extra-configs --no-cpu-baseline --no-gpu-baseline > gpurun_out/r05n_trace.log 2>&1
python tools/trace_overlap.py gpurun_out/r05n_trace 7 8 --dump gpurun_out/r05n_dispatch_graph.txt
rm -rf gpurun_out/r05n_trace
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_bf16.py -x -q > gpurun_out/r05n_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r05n_tests.log
