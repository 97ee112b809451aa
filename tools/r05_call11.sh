#!/bin/bash
# round 5, GPU call 11: counters + kernel stats of the tree (fp32 headline and the bf16 config), the full bench line, the full GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash tools/profile_bench.sh r05k > gpurun_out/r05k_profile.log 2>&1; echo "profile rc=$?"
VXM_PROFILE_SUFFIX=_bf16 bash tools/profile_bench.sh r05k --config dense_bf16 > gpurun_out/r05k_profile_bf16.log 2>&1; echo "profile bf16 rc=$?"
ls gpurun_out | grep r05k
head -12 gpurun_out/r05k_rocprof_kernel_stats_serial.csv | cut -c1-120
