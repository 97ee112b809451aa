#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-gpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B $EXTRA > gpurun_out/r05g_$tag.json 2> gpurun_out/r05g_$tag.err; python - $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/r05g_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
    print("%-28s value %.2f  ms %.3f  host %.2f  replays %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["submission"]["graph_replays"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
EXTRA=""
run eager VXM_GRAPH=0
run graph_default VXM_GRAPH=1
run graph_q1 VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run graph_q2 VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run graph_q3 VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=3
run graph_q8 VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run graph_q2b VXM_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run eager2 VXM_GRAPH=0
