#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest -q -x -m gpu -s tests/test_gpu_parity.py -k "other_pooling or pooling_kernels" 2>&1 | grep -E "gate|passed|failed|Error|error" | tail -30
for pc in 0 1 0 1; do
  echo "== VXM_S3_PC=$pc"; VXM_S3_PC=$pc python tools/s3_bench.py --iters 10 --only " fwd " 2>&1 | grep -E "^rem[12] "
done
