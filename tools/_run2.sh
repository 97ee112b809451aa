set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03b; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_s3.py -q -m gpu -s -k "forward_vs_fp64 or fused_mask or scale_invariance or other_kernel" > $O/s3_direct.log 2>&1; echo "s3 direct rc=$?"; tail -3 $O/s3_direct.log
timeout 200 python tools/s3_bench.py --iters 5 > $O/s3_bench_default.log 2>&1; echo "bench default rc=$?"
VXM_S3_CB=2 timeout 200 python tools/s3_bench.py --iters 5 > $O/s3_bench_cb2.log 2>&1; echo "bench cb2 rc=$?"
VXM_S3_NCT=1 timeout 200 python tools/s3_bench.py --iters 5 --only "32" > $O/s3_bench_nct1.log 2>&1; echo "bench nct1 rc=$?"
bash tools/pmc_run.sh $O/pmc k_s3_conv -- python tools/s3_bench.py --iters 3 --only "rem" > /dev/null 2>&1; echo "pmc rc=$?"
cat $O/s3_bench_default.log $O/s3_bench_cb2.log $O/s3_bench_nct1.log | grep -v amdgpu
cat $O/pmc/summary.txt
