#!/bin/bash
# round 5, GPU call 5: graph vs eager same-box A/B after the glue changes, bf16 graph / eager / one stream, dispatch lists, tests touched since call 4,
# the torch-ROCm leg (ATen convolutions) with a long deadline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-gpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B $EXTRA > gpurun_out/r05e_$tag.json 2> gpurun_out/r05e_$tag.err; python - $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/r05e_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
    print("%-28s value %.2f  ms %.3f  host %.2f  replays %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["submission"]["graph_replays"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
EXTRA=""
run graph1 VXM_GRAPH=1; run eager1 VXM_GRAPH=0; run graph2 VXM_GRAPH=1; run eager2 VXM_GRAPH=0; run graph3 VXM_GRAPH=1; run eager3 VXM_GRAPH=0
EXTRA="--config dense_bf16"
run bf16_graph VXM_GRAPH=1; run bf16_eager VXM_GRAPH=0; run bf16_eager_1stream VXM_GRAPH=0 VXM_NO_OVERLAP=1; run bf16_graph_1stream VXM_GRAPH=1 VXM_NO_OVERLAP=1
for mode in 1 0; do
  rm -rf gpurun_out/r05e_trace_$mode
  VXM_GRAPH=$mode timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05e_trace_$mode -- python bench.py --steps 6 --warmup 4 --no-extra-configs --no-cpu-baseline --no-gpu-baseline > gpurun_out/r05e_trace_$mode.log 2>&1
  python tools/trace_overlap.py gpurun_out/r05e_trace_$mode 7 8 --dump gpurun_out/r05e_dispatch_mode$mode.txt
  rm -rf gpurun_out/r05e_trace_$mode
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py -x -q -s -k "heavy_tailed or umbrella or range_report or train_and_register or graph_replay" > gpurun_out/r05e_tests.log 2>&1; echo "tests rc=$?"
grep -n "noise pair\|heavy tails\|passed\|failed\|^E " gpurun_out/r05e_tests.log | head -20
timeout 1000 python bench.py --torch-rocm-baseline-worker --shape 160,192,224 --int-steps 7 > gpurun_out/r05e_torch_rocm_baseline.json 2> gpurun_out/r05e_torch_rocm_baseline.err; echo "torch-rocm rc=$?"
cat gpurun_out/r05e_torch_rocm_baseline.json; tail -3 gpurun_out/r05e_torch_rocm_baseline.err
