set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03n; mkdir -p $O
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench$i.json 2> $O/bench$i.err; echo "bench rc=$?"
done
python - <<'PY'
import json
for i in (1,2):
    d=json.loads(open("gpurun_out/r03n/bench%d.json"%i).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["measured_in"])
    for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:4]:
        print("   %-40s %.3f ms/step  launches %.1f  %s" % (k, v["ms_per_step"], v["launches_per_step"], ("%.1f TF"%v["tflops"]) if "tflops" in v else ""))
PY
rocm-smi --showclocks --showpower 2>/dev/null | head -20
