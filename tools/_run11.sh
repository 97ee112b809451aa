set -u
cd "$GRAFT_REPO_ROOT"
bash tools/profile_bench.sh r03k > gpurun_out/r03k_profile.log 2>&1; echo "profile rc=$?"
ls gpurun_out | grep r03k
python - <<'PY'
import csv
rows=list(csv.reader(open("gpurun_out/r03k_rocprof_kernel_stats.csv")))
for r in rows[:22]: print(r[:4], r[-1] if len(r)>4 else "")
PY
