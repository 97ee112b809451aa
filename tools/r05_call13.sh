#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-gpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B $EXTRA > gpurun_out/r05m_$tag.json 2> gpurun_out/r05m_$tag.err; python - $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/r05m_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
    print("%-24s value %.2f  ms %.3f  host %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
EXTRA=""
run warm VXM_GRAPH=0
run graph_before VXM_GRAPH=1
run graph_after VXM_GRAPH=1 VXM_DW_ORDER=after
run eager_before VXM_GRAPH=0
run eager_after VXM_GRAPH=0 VXM_DW_ORDER=after
run graph_before2 VXM_GRAPH=1
run graph_after2 VXM_GRAPH=1 VXM_DW_ORDER=after
run eager_before2 VXM_GRAPH=0
run eager_after2 VXM_GRAPH=0 VXM_DW_ORDER=after
rm -rf gpurun_out/r05m_trace gpurun_out/r05m_dispatch_graph_after.txt
VXM_GRAPH=1 VXM_DW_ORDER=after timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05m_trace -- python bench.py --steps 6 --warmup 4 --no-extra-configs --no-cpu-baseline --no-gpu-baseline > gpurun_out/r05m_trace.log 2>&1
python tools/trace_overlap.py gpurun_out/r05m_trace 7 8 --dump gpurun_out/r05m_dispatch_graph_after.txt
rm -rf gpurun_out/r05m_trace
