#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-gpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B $EXTRA > gpurun_out/r05i_$tag.json 2> gpurun_out/r05i_$tag.err; python - $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/r05i_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernels"]
    print("%-24s value %.2f  ms %.3f  host %.2f | s3p %.3f s3_conv<1> %.3f s3_conv<2> %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"],
          k.get("k_s3p_conv<1,2>", {}).get("ms_per_step", 0), k.get("k_s3_conv<1,8,1,2>", {}).get("ms_per_step", 0), k.get("k_s3_conv<2,8,1,2>", {}).get("ms_per_step", 0)))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
EXTRA=""
run warm VXM_GRAPH=0
run eager_snake VXM_GRAPH=0 VXM_S3_SNAKE=1
run eager_plain VXM_GRAPH=0 VXM_S3_SNAKE=0
run graph_snake VXM_GRAPH=1 VXM_S3_SNAKE=1
run graph_plain VXM_GRAPH=1 VXM_S3_SNAKE=0
run eager_snake2 VXM_GRAPH=0 VXM_S3_SNAKE=1
run eager_plain2 VXM_GRAPH=0 VXM_S3_SNAKE=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py -x -q -k "graph_replay or channel_blocked or full_size_train_step_vs_oracle_noise or unet_vs_oracle or vxm_dense_golden or umbrella" > gpurun_out/r05i_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r05i_tests.log
