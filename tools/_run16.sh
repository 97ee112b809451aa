set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03p; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -s --maxfail=25 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED" $O/gpu_tests.log | tail -8
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
bash tools/profile_bench.sh r03p > $O/profile.log 2>&1; echo "profile rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03p/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"])
for k,v in d.get("extra_configs",{}).items(): print("  extra", k, {kk:vv for kk,vv in v.items() if kk in ("value","ms_per_step","error")})
print(d.get("cpu_baseline",{}).get("value"))
PY
