#!/bin/bash
# round 5, GPU call 2: why is the replayed step slower than the eager one?  graph / eager x two streams / one stream, runtime switches, traces
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-gpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B > gpurun_out/r05b_$tag.json 2> gpurun_out/r05b_$tag.err; python - $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/r05b_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
    print("%-28s value %.2f  ms %.3f  host %.2f  replays %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["submission"]["graph_replays"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
run graph            VXM_GRAPH=1
run eager            VXM_GRAPH=0
run graph_1stream    VXM_GRAPH=1 VXM_NO_OVERLAP=1
run eager_1stream    VXM_GRAPH=0 VXM_NO_OVERLAP=1
run graph_devkernarg VXM_GRAPH=1 HIP_FORCE_DEV_KERNARG=1
run graph_nodevkarg  VXM_GRAPH=1 HIP_FORCE_DEV_KERNARG=0
run graph_pktcap0    VXM_GRAPH=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run graph_pktcap1    VXM_GRAPH=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run graph_again      VXM_GRAPH=1
for mode in 1 0; do
  rm -rf gpurun_out/r05b_trace_$mode
  VXM_GRAPH=$mode timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05b_trace_$mode -- python bench.py --steps 6 --warmup 4 --no-extra-configs --no-cpu-baseline --no-gpu-baseline > gpurun_out/r05b_trace_$mode.log 2>&1
  echo "== trace VXM_GRAPH=$mode"; python tools/trace_overlap.py gpurun_out/r05b_trace_$mode 6 8
  rm -rf gpurun_out/r05b_trace_$mode
done
