#!/bin/bash
# LDS counters of the weight-gradient and forward kernels on the 420 k-row level (evidence for HISTORY.md 4b).
export TMPDIR=/tmp; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04q_pmc_dw.txt; : > $OUT
for pass in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  D=/tmp/pmc_$RANDOM; rm -rf $D
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $D -o p -- python $ROOT/scripts/bench_conv.py --cases 16x16,8x8 --iters 10 > /dev/null 2>&1)
  F=$(find $D -name '*counter_collection.csv' | head -1)
  echo "== pass: $pass" >> $OUT
  [ -n "$F" ] && python - >> $OUT <<PY
import csv, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('$F')):
    n = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('void ', ''))
    if 'k_conv_dw' in n or 'k_conv_fwd' in n:
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n, cs in sorted(agg.items()):
    print('%-44s' % n[:44], '  '.join('%s=%.3g (n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
PY
done
cat $OUT
