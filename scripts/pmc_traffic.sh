#!/bin/bash
# HBM traffic of the bench's kernels from the L2 memory-side counters: two separate --pmc passes (FETCH_SIZE uses 3
# TCC counters, WRITE_SIZE 2: they do not fit one pass).  Writes gpurun_out/pmc_traffic.json (per-kernel sums and
# launch counts); copy it to profiles/ to have bench.py report `roofline.traffic`.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_traffic
rm -rf $OUT
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT.fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT.write.log 2>&1
python3 - <<PY
import csv, glob, json, collections
res = collections.defaultdict(lambda: dict(launches=0, FETCH_SIZE=0.0, WRITE_SIZE=0.0))
for sub, ctr in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    f = glob.glob('$OUT/%s/*/*counter_collection.csv' % sub)[0]
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != ctr: continue
        k = r['Kernel_Name']
        fam = ('conv3x3_igemm_p' if 'conv3x3_igemm_p' in k else 'conv3x3_igemm_tile' if 'conv3x3_igemm_tile' in k else
               'conv3x3_wgrad' if 'conv3x3_wgradI' in k else
               'bn_bwd' if 'bn_bwd' in k and 'finalize' not in k else
               'bn_act_pool_fwd' if ('bn_act_pool' in k or 'bn_pool_rows_fwd' in k) else
               'conv3x3_first_fwd' if 'first_fwd' in k else 'conv3x3_first_wgrad' if 'first_wgrad' in k else None)
        if fam is None: continue
        res[fam][ctr] += float(r['Counter_Value'])
        if ctr == 'FETCH_SIZE': res[fam]['launches'] += 1
json.dump(res, open('gpurun_out/pmc_traffic.json', 'w'), indent=1)
for k, v in res.items(): print(k, v)
PY
