set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv_block_vs_oracle or full_size_conv_adjoint or bitwise_deterministic or multi_feature or vxm_dense_golden" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED|Error" $O/tests.log | tail -8
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
VXM_FEWCH=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/bench_off.json 2> $O/bench_off.err; echo "bench off rc=$?"
python - <<'PY'
import json
for f in ("bench","bench_off"):
    d=json.loads(open("gpurun_out/r03o/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["measured_in"][:60])
    for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"]):
        if "vec<1>" in k or "fewch" in k or "dma" in k: print("   %-40s %.3f ms/step  launches %.1f  %s" % (k, v["ms_per_step"], v["launches_per_step"], ("%.1f TF"%v["tflops"]) if "tflops" in v else ""))
PY
