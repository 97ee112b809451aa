#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r05w_bench.json 2> gpurun_out/r05w_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05w_bench.json") if l.startswith("{")][-1])
print("value %.2f ms %.3f host %.2f" % (d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"]), d["submission"])
print("roofline", {k: v for k, v in d["roofline"].items() if k not in ("peak_note", "traffic_unit")})
print("traffic_unit", d["roofline"]["traffic_unit"][:200])
print("conv_mfma_util", d.get("conv_mfma_util", {}).get("time_weighted"), d.get("conv_mfma_util", {}).get("covered_fraction_of_conv_time"))
print("st+vecint", d["spatial_transformer_plus_vecint"])
for k, v in d.get("extra_configs", {}).items():
    print("   ", k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("value", "ms_per_step", "ms_per_pair", "host_enqueue_ms_per_step", "error")}, (v.get("spatial_transformer_plus_vecint") or {}).get("frac_of_hbm_peak"))
print("gpu_baseline", d.get("gpu_baseline")); print("cpu_baseline", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k in ("value", "cores", "kind")})
PY
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05w_gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -4 gpurun_out/r05w_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
