#!/bin/bash
# usage: pmc_conv.sh <layer> <kind>   -> per-kernel counter sums for the conv kernel of that layer
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
L=$1; K=$2
OUT=gpurun_out/pmc_${L}_${K}
rm -rf $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 scripts/conv_micro.py $L $K 3 > $OUT.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/b -- python3 scripts/conv_micro.py $L $K 3 >> $OUT.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in 'ab':
    f = glob.glob('$OUT/%s/*/*counter_collection.csv' % sub)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv3x3' not in k or 'reduce' in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
    for k, d in acc.items():
        print(k[:70])
        for c, v in d.items(): print('   %-28s %.4g' % (c, v))
PY
tail -1 $OUT.log
