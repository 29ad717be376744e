#!/bin/bash
# MFMA utilisation evidence per kernel family (BASELINE.json north_star: "MFMA utilisation for the conv, GEMM and
# classifier kernels"): two --pmc passes over the bench (serialised by the profiler), summed per family.
#   mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs)   (the gfx94x MfmaUtil formula)
#   valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA,  lds conflict cycles per LDS instruction
# usage: scripts/pmc_mfma.sh [OUT.json]   (writes gpurun_out/pmc_mfma.json; copy to profiles/)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_mfma
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
python3 - "${1:-gpurun_out/pmc_mfma.json}" <<PY
import csv, glob, json, collections, re, sys
def family(k):
    if 'conv3x3_igemm_tile' in k: return 'conv3x3_igemm_tile'
    if 'conv3x3_igemm_p' in k: return 'conv3x3_igemm_p'
    if 'conv3x3_wgradI' in k or 'conv3x3_wgrad<' in k:
        deep = re.search(r'Li32ELi64E|, 32, 64,', k) is not None
        return 'conv3x3_wgrad_deep(Cin>=32,Cout>=64)' if deep else 'conv3x3_wgrad_shallow'
    if 'first_fwd' in k: return 'conv3x3_first_fwd'
    if 'first_wgrad' in k and 'reduce' not in k: return 'conv3x3_first_wgrad'
    if 'gemm_bf16' in k:
        rich = re.search(r'Lb1EEEv|, true>\(', k) is not None
        return 'gemm_bf16_rich(vqa_fc_bce)' if rich else 'gemm_bf16'
    if 'gru_step_fwd' in k: return 'gru_step_fwd_k'
    if 'gru_step_bwd' in k: return 'gru_step_bwd_k'
    return None
res = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in 'ab':
    f = glob.glob('$OUT/%s/*/*counter_collection.csv' % sub)[0]
    first = {}
    for r in csv.DictReader(open(f)):
        fam = family(r['Kernel_Name'])
        if fam is None: continue
        res[fam][r['Counter_Name']] += float(r['Counter_Value'])
        key = (sub, fam, r['Counter_Name'])
        first[key] = first.get(key, 0) + 1
    for (s, fam, c), n in first.items():
        res[fam]['launches_' + s] = max(res[fam]['launches_' + s], n)
out = {}
for fam, d in sorted(res.items()):
    gui = d.get('GRBM_GUI_ACTIVE', 0.0)
    o = dict(launches=int(d['launches_a']), counters={k: v for k, v in d.items() if not k.startswith('launches')})
    if gui: o['mfma_busy_frac'] = round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 256 * 4), 4)
    if d.get('SQ_BUSY_CYCLES'): o['mfma_busy_over_sq_busy'] = round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_BUSY_CYCLES'], 4)
    if d.get('SQ_INSTS_MFMA'): o['valu_per_mfma'] = round(d['SQ_INSTS_VALU'] / d['SQ_INSTS_MFMA'], 2)
    if d.get('SQ_INSTS_LDS'): o['lds_conflict_cycles_per_lds_inst'] = round(d['SQ_LDS_BANK_CONFLICT'] / d['SQ_INSTS_LDS'], 3)
    if d.get('SQ_WAVE_CYCLES'): o['valu_active_over_wave_cycles'] = round(d['SQ_ACTIVE_INST_VALU'] / d['SQ_WAVE_CYCLES'], 4)
    out[fam] = o
    print(fam, {k: v for k, v in o.items() if k != 'counters'})
json.dump(out, open(sys.argv[1], 'w'), indent=1)
PY
