cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for op in fwd bww; do for lay in planar blocked; do
  for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pm; timeout 100 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pm -- python tools/blk_counters.py $op $lay > /tmp/pm.log 2>&1
    python tools/rocprof_summary.py pmc /tmp/pm /tmp/pm.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/pm.json'))
for k,v in d.items():
    if ('k_s3p_conv' in k or 'k_s3_bwd_weight' in k): print('$op $lay', k, {c: round(x['mean'],1) for c,x in v.items() if isinstance(x,dict)}, v.get('duration_us_under_pmc'))
" 2>&1 | tail -2
  done
done; done
