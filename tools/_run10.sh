set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_s3.py -q -m gpu -s -k "forward_vs_fp64 or fused_mask or scale_invariance or other_kernel or through_the_dispatcher" > $O/s3.log 2>&1; echo "s3 rc=$?"; grep -E "passed|failed|FAILED" $O/s3.log | tail -5
timeout 200 python tools/s3_bench.py --iters 5 --only "rem" 2>&1 | grep -v amdgpu | tee $O/s3_bench.log
VXM_S3_ROWS32=2 timeout 200 python tools/s3_bench.py --iters 5 --only "->32" 2>&1 | grep -v amdgpu | tee $O/s3_bench_rows2.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
VXM_S3_ROWS32=2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/bench_rows2.json 2> $O/bench_rows2.err; echo "bench rows2 rc=$?"
python - <<'PY'
import json
for f in ("bench","bench_rows2"):
    d=json.loads(open("gpurun_out/r03j/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"])
    for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:6]:
        print("   %-40s %.3f ms/step  launches %.1f  %s" % (k, v["ms_per_step"], v["launches_per_step"], ("%.1f TF"%v["tflops"]) if "tflops" in v else ""))
PY
