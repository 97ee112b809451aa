#!/bin/bash
# round 5, GPU call 6: reductions of the weight gradients on a third stream: A/B graph / eager, dispatch lists, tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-gpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B $EXTRA > gpurun_out/r05f_$tag.json 2> gpurun_out/r05f_$tag.err; python - $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/r05f_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
    print("%-28s value %.2f  ms %.3f  host %.2f  replays %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["submission"]["graph_replays"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
EXTRA=""
run graph1 VXM_GRAPH=1; run eager1 VXM_GRAPH=0; run graph2 VXM_GRAPH=1; run eager2 VXM_GRAPH=0
for mode in 1 0; do
  rm -rf gpurun_out/r05f_trace_$mode gpurun_out/r05f_dispatch_mode$mode.txt
  VXM_GRAPH=$mode timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05f_trace_$mode -- python bench.py --steps 6 --warmup 4 --no-extra-configs --no-cpu-baseline --no-gpu-baseline > gpurun_out/r05f_trace_$mode.log 2>&1
  python tools/trace_overlap.py gpurun_out/r05f_trace_$mode 8 8 --dump gpurun_out/r05f_dispatch_mode$mode.txt
  rm -rf gpurun_out/r05f_trace_$mode
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py -x -q -k "umbrella or graph_replay or two_rank or channel_blocked or full_size_train_step_vs_oracle_noise or unet_vs_oracle or vxm_dense_golden or bwd_weight_bitwise or train_and_register" > gpurun_out/r05f_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r05f_tests.log
