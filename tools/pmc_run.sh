#!/bin/bash
# GPU box helper: kernel trace + two PMC passes (separate, kernel-trace only) of an arbitrary command; prints tools/pmc_summary.py
# usage: tools/pmc_run.sh OUTDIR FILTER -- command...
set -u
OUT=$1; FLT=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- "$@" > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p2 -- "$@" > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p3 -- "$@" > $OUT/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_MFMA --output-format csv -d $OUT/p4 -- "$@" > $OUT/p4.log 2>&1
PMC_PER_DISPATCH=1 python tools/pmc_summary.py $OUT "$FLT" > $OUT/summary.txt 2>&1
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
