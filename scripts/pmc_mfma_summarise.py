"""Summarise the two --pmc passes of scripts/pmc_mfma.sh per kernel family: python3 scripts/pmc_mfma_summarise.py OUT.json DIR"""

import csv, glob, json, collections, re, sys
OUT = sys.argv[2]
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
    f = glob.glob('%s/%s/*/*counter_collection.csv' % (OUT, sub))[0]
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
    # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (it reads 8 x the kernel's cycles); a 16x16x32 bf16 MFMA is counted
    # as two 16-cycle operations by SQ_INSTS_MFMA / SQ_VALU_MFMA_BUSY_CYCLES (32 pipe cycles per instruction)
    if gui: o['mfma_busy_frac'] = round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui / 8 * 256 * 4), 4)
    if d.get('SQ_BUSY_CYCLES'): o['mfma_busy_over_sq_busy'] = round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_BUSY_CYCLES'], 4)
    if d.get('SQ_INSTS_MFMA'): o['valu_per_mfma'] = round(d['SQ_INSTS_VALU'] / d['SQ_INSTS_MFMA'], 2)
    if d.get('SQ_INSTS_LDS'): o['lds_conflict_cycles_per_lds_inst'] = round(d['SQ_LDS_BANK_CONFLICT'] / d['SQ_INSTS_LDS'], 3)
    if d.get('SQ_WAVE_CYCLES'): o['valu_active_over_wave_cycles'] = round(d['SQ_ACTIVE_INST_VALU'] / d['SQ_WAVE_CYCLES'], 4)
    out[fam] = o
    print(fam, {k: v for k, v in o.items() if k != 'counters'})
json.dump(out, open(sys.argv[1], 'w'), indent=1)
