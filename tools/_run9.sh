set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03i; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -s --maxfail=25 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED" $O/gpu_tests.log | tail -8
timeout 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03i/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:14]:
    print("   %-40s %.3f ms/step  launches %.1f  %s" % (k, v["ms_per_step"], v["launches_per_step"], ("%.1f TF"%v["tflops"]) if "tflops" in v else ("%.0f GB/s"%v.get("gbs",0))))
for k,v in d.get("extra_configs",{}).items(): print("  extra", k, {kk:vv for kk,vv in v.items() if kk in ("value","ms_per_step","error")})
PY
