set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
timeout 300 python -m pytest tests/test_gpu_s3.py -q -m gpu -s -k "forward_vs_fp64 or fused_mask or scale_invariance" > $O/s3_direct.log 2>&1; echo "s3 direct rc=$?" 
timeout 200 python tools/s3_bench.py --iters 5 --json $O/s3_bench_default.json > $O/s3_bench_default.log 2>&1; echo "bench default rc=$?"
VXM_S3_CB=2 timeout 200 python tools/s3_bench.py --iters 5 --json $O/s3_bench_cb2.json > $O/s3_bench_cb2.log 2>&1; echo "bench cb2 rc=$?"
VXM_S3_NCT=1 timeout 200 python tools/s3_bench.py --iters 5 --only "->32" > $O/s3_bench_nct1.log 2>&1; echo "bench nct1 rc=$?"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_split.json 2> $O/bench_split.err; echo "bench split rc=$?"
VXM_FP32_ENGINE=native timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/bench_native.json 2> $O/bench_native.err; echo "bench native rc=$?"
timeout 1500 python -m pytest tests -q -m gpu -s --maxfail=25 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"
tail -5 $O/gpu_tests.log
cat $O/s3_bench_default.log
