#!/bin/bash
# round 5, GPU call 3: far-pass merge (VecInt tests), heavy-tailed step on three engines, graph tests, bf16 config kernel stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_parity.py -x -q -s -k "graph or two_rank or bench_two or adam_device or vecint or heavy_tailed" > gpurun_out/r05c_tests.log 2>&1; echo "tests rc=$?"
grep -n "heavy tails\|passed\|failed\|Error\|assert" gpurun_out/r05c_tests.log | head -30
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-gpu-baseline > gpurun_out/r05c_bench.json 2> gpurun_out/r05c_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05c_bench.json") if l.startswith("{")][-1])
print("value %.2f ms %.3f host %.2f" % (d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"]), d["roofline"]["measured_in"])
for k in ("vecint_bwd", "vecint_fwd", "k_s3_bwd_weight<2>", "resize3d_bwd"):
    print("  ", k, d["kernels"][k])
print(d["spatial_transformer_plus_vecint"])
PY
rm -rf gpurun_out/r05c_bf16prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05c_bf16prof -- python bench.py --config dense_bf16 --steps 6 --warmup 4 --no-extra-configs --no-cpu-baseline --no-gpu-baseline > gpurun_out/r05c_bf16prof.log 2>&1
VXM_GRAPH=0 VXM_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05c_bf16prof_serial -- python bench.py --config dense_bf16 --steps 6 --warmup 4 --no-extra-configs --no-cpu-baseline --no-gpu-baseline > gpurun_out/r05c_bf16prof_serial.log 2>&1
python tools/rocprof_summary.py stats gpurun_out/r05c_bf16prof_serial gpurun_out/r05c_bf16_kernel_stats_serial.csv
python tools/trace_overlap.py gpurun_out/r05c_bf16prof 5 8
grep -h '^{' gpurun_out/r05c_bf16prof.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 under rocprof: value %.1f ms %.3f'%(d['value'], d['ms_per_step']))"
head -30 gpurun_out/r05c_bf16_kernel_stats_serial.csv | cut -c1-150
rm -rf gpurun_out/r05c_bf16prof gpurun_out/r05c_bf16prof_serial
