set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03q; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_s3.py tests/test_gpu_parity.py -q -m gpu -k "forward_vs_fp64 or fused_mask or few_channel" > $O/t.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/t.log | tail -2
VXM_S3_TWOPASS=1 timeout 300 python -m pytest tests/test_gpu_s3.py -q -m gpu -k "forward_vs_fp64 or fused_mask" > $O/t2.log 2>&1; echo "twopass tests rc=$?"; grep -E "passed|failed" $O/t2.log | tail -2
timeout 200 python tools/s3_bench.py --iters 5 --only "32" 2>&1 | grep -v amdgpu | grep -v weight | tee $O/b1.log
VXM_S3_TWOPASS=1 timeout 200 python tools/s3_bench.py --iters 5 --only "32" 2>&1 | grep -v amdgpu | grep -v weight | tee $O/b2.log
