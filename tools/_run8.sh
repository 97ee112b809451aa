set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "vecint or vxm_dense_golden or full_size_train_step_vs_oracle_noise" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED" $O/tests.log | tail -5
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03h/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k in ("vecint_bwd","vecint_fwd","warp3d_fwd","warp3d_bwd"):
    v=d["kernels"].get(k)
    if v: print("   %-20s %.3f ms/step %s" % (k, v["ms_per_step"], ("%.0f GB/s"%v.get("gbs",0))))
PY
