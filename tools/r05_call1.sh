#!/bin/bash
# round 5, GPU call 1: graph-step tests, the default bench (graph submission), the same with eager submission
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_graph.py -x -q -s > gpurun_out/r05a_graph_tests.log 2>&1; echo "graph tests rc=$?"
tail -15 gpurun_out/r05a_graph_tests.log
timeout 900 python bench.py > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; echo "bench rc=$?"
VXM_GRAPH=0 timeout 600 python bench.py --no-cpu-baseline --no-gpu-baseline > gpurun_out/r05a_bench_eager.json 2> gpurun_out/r05a_bench_eager.err; echo "eager bench rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r05a_bench.json", "gpurun_out/r05a_bench_eager.json"):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "value %.2f ms %.3f host %.2f" % (d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"]), d.get("submission"))
    for k, v in d.get("extra_configs", {}).items():
        print("   ", k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("value", "ms_per_step", "ms_per_pair", "host_enqueue_ms_per_step", "submission", "error")})
    print("    gpu_baseline", d.get("gpu_baseline")); print("    cpu_baseline", (d.get("cpu_baseline") or {}).get("value"))
PY
tail -5 gpurun_out/r05a_bench.err gpurun_out/r05a_bench_eager.err
