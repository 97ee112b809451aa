#!/bin/bash
# round 5, GPU call 4: glue kernels (wmax, bww reduce, side-stream prepack), bf16 side stream, range probe; tests + full bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_s3.py tests/test_gpu_bf16.py -x -q -s > gpurun_out/r05d_tests.log 2>&1; echo "tests rc=$?"
grep -n "noise pair\|heavy tails\|passed\|failed\|^E " gpurun_out/r05d_tests.log | head -30
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "heavy_tailed or unet_vs_oracle or vxm_dense_golden or full_size_train_step_vs_oracle_noise or channel_blocked" > gpurun_out/r05d_tests2.log 2>&1; echo "tests2 rc=$?"
grep -n "heavy tails\|passed\|failed\|^E " gpurun_out/r05d_tests2.log | head -30
timeout 900 python bench.py --no-cpu-baseline --no-gpu-baseline > gpurun_out/r05d_bench.json 2> gpurun_out/r05d_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05d_bench.json") if l.startswith("{")][-1])
print("value %.2f ms %.3f host %.2f" % (d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"]), d["roofline"]["measured_in"], d["submission"])
tot = 0
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    tot += v["ms_per_step"]
print("sum of per-kernel regions %.3f" % tot)
for k, v in d.get("extra_configs", {}).items():
    print("   ", k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("value", "ms_per_step", "ms_per_pair", "host_enqueue_ms_per_step", "error")})
PY
tail -3 gpurun_out/r05d_bench.err
