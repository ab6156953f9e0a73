#!/bin/bash
# Counters of the row-contiguous-gather kernel (k_conv_fwd_rc) and of k_conv_fwd_w on the 366 k-row level.  Usage: scripts/pmc_rc.sh [TAG]
TAG=${1:-r06q_pmc_rc}
export TMPDIR=/tmp; ROOT=$(pwd); OUT=$ROOT/gpurun_out/${TAG}.txt; : > $OUT
for rg in 0 1; do
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_TA_BUSY GRBM_GUI_ACTIVE" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES" ; do
  D=/tmp/pmc_$RANDOM; rm -rf $D
  (cd /tmp && SGNN_TUNE=conv_row_gather=$rg timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $D -o p -- python $ROOT/scripts/bench_epi.py --iters 5 > /dev/null 2>&1)
  F=$(find $D -name '*counter_collection.csv' | head -1)
  echo "== conv_row_gather=$rg pass: $pass" >> $OUT
  [ -n "$F" ] && python - >> $OUT <<PY
import csv, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('$F')):
    n = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('void ', ''))
    if 'k_conv_fwd_w<16' in n or 'k_conv_fwd_rc' in n:
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n, cs in sorted(agg.items()):
    print('%-32s' % n[:32], '  '.join('%s=%.4g (n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
PY
done; done
cat $OUT
