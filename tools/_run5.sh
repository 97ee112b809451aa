set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03e; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_s3.py -q -m gpu -s -k "backward_weight" > $O/s3_bw.log 2>&1; echo "s3 bw rc=$?"; grep -E "passed|failed|Error|error" $O/s3_bw.log | head -5
timeout 200 python tools/s3_bench.py --iters 5 --only "bwd-weight" 2>&1 | grep -v amdgpu | tee $O/s3_bench.log
bash tools/pmc_run.sh $O/pmc k_s3_bwd_weight -- python tools/s3_bench.py --iters 3 --only "rem1 bwd-weight" > /dev/null 2>&1; grep -E "MFMA_BUSY|GRBM|WAIT|ACTIVE_INST_ANY|WAVE_CYCLES|avg|INSTS_VALU|LDS_IDX" $O/pmc/summary.txt | grep -v "per dispatch"
