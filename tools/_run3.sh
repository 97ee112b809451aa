set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_s3.py -q -m gpu -s -k "backward_weight" > $O/s3_bw.log 2>&1; echo "s3 bw rc=$?"; grep -E "rel-L2|passed|failed|Error|error" $O/s3_bw.log | head -20
timeout 200 python tools/s3_bench.py --iters 5 --only "bwd-weight" > $O/s3_bench_bw.log 2>&1; echo "bench rc=$?"; grep -v amdgpu $O/s3_bench_bw.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/bench_split.json 2> $O/bench_split.err; echo "bench split rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03c/bench_split.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:12]:
    print("   %-40s %.3f ms/step  launches %.1f  %s" % (k, v["ms_per_step"], v["launches_per_step"], ("%.1f TF"%v["tflops"]) if "tflops" in v else ""))
PY
bash tools/pmc_run.sh $O/pmc k_s3_bwd_weight -- python tools/s3_bench.py --iters 3 --only "rem1 bwd-weight" > /dev/null 2>&1; cat $O/pmc/summary.txt
