#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-gpu-baseline --config dense_bf16"
run() { tag=$1; shift; env "$@" timeout 300 $B > gpurun_out/r05j_$tag.json 2> gpurun_out/r05j_$tag.err; python - $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/r05j_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
    print("%-24s value %.2f  ms %.3f  host %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
run warm VXM_GRAPH=0
run eager_snake VXM_GRAPH=0 VXM_S3_SNAKE=1
run eager_plain VXM_GRAPH=0 VXM_S3_SNAKE=0
run graph_snake VXM_GRAPH=1 VXM_S3_SNAKE=1
run graph_plain VXM_GRAPH=1 VXM_S3_SNAKE=0
run eager_snake2 VXM_GRAPH=0 VXM_S3_SNAKE=1
run eager_plain2 VXM_GRAPH=0 VXM_S3_SNAKE=0
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q > gpurun_out/r05j_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r05j_tests.log
