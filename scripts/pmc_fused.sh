#!/bin/bash
# Counters of the fused backward kernel (and the two kernels it replaces) on the 366 k-row level.  Usage: scripts/pmc_fused.sh [TAG]
TAG=${1:-r06_pmc_fused}
export TMPDIR=/tmp; ROOT=$(pwd); OUT=$ROOT/gpurun_out/${TAG}.txt; : > $OUT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES" \
            "SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU" ; do
  D=/tmp/pmc_$RANDOM; rm -rf $D
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $D -o p -- python $ROOT/scripts/bench_bwd_fused.py --iters 5 > /dev/null 2>&1)
  F=$(find $D -name '*counter_collection.csv' | head -1)
  echo "== pass: $pass" >> $OUT
  [ -n "$F" ] && python - >> $OUT <<PY
import csv, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('$F')):
    n = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('void ', ''))
    if 'k_conv_dw<' in n or 'k_conv_fwd_w' in n or 'k_conv_bwd_fused' in n:
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n, cs in sorted(agg.items()):
    print('%-40s' % n[:40], '  '.join('%s=%.4g (n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
PY
done
cat $OUT
