set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03m; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_s3.py -q -m gpu -s -k "backward_weight" > $O/s3_bw.log 2>&1; echo "s3 bw rc=$?"; grep -E "rel-L2|passed|failed" $O/s3_bw.log | tail -6
timeout 600 python tools/bw_accuracy.py 2>&1 | grep -v amdgpu | tee $O/bw_accuracy.log
timeout 200 python tools/s3_bench.py --iters 5 --only "bwd-weight" 2>&1 | grep -v amdgpu | tee $O/s3_bench.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03m/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:4]:
    print("   %-40s %.3f ms/step  launches %.1f  %s" % (k, v["ms_per_step"], v["launches_per_step"], ("%.1f TF"%v["tflops"]) if "tflops" in v else ""))
PY
