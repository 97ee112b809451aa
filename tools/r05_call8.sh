#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-gpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B $EXTRA > gpurun_out/r05h_$tag.json 2> gpurun_out/r05h_$tag.err; python - $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/r05h_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
    print("%-28s value %.2f  ms %.3f  host %.2f  replays %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["submission"]["graph_replays"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
EXTRA=""
run graph_q8_cold VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run eager VXM_GRAPH=0
run graph_q4 VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run graph_q6 VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=6
run graph_q8 VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run graph_q12 VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=12
run graph_q16 VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=16
run graph_q32 VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=32
run graph_q8b VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run eager2 VXM_GRAPH=0
EXTRA="--config dense_bf16"
run bf16_graph_q8 VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run bf16_eager VXM_GRAPH=0
EXTRA="--batch-per-gpu 4 --steps 6 --warmup 4"
run b4_graph_q8 VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run b4_graph VXM_GRAPH=1
run b4_eager VXM_GRAPH=0
