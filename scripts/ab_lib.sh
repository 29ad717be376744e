#!/bin/bash
# Same-box A/B of two builds of the library: scripts/ab_lib.sh <base.so> <new.so> [rounds]
# alternates bench.py --breakdown runs and prints step time plus the per-entry kernel totals of each.
base=$1; new=$2; rounds=${3:-2}
for i in $(seq $rounds); do
  for tag in base new; do
    lib=$base; [ $tag = new ] && lib=$new
    VQA_HIP_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --breakdown > gpurun_out/ab_${tag}_$i.log 2>&1 || { echo "$tag FAILED"; tail -3 gpurun_out/ab_${tag}_$i.log; exit 1; }
    python - $tag $i <<PY
import json, sys, re
tag, i = sys.argv[1], sys.argv[2]
txt = open(f'gpurun_out/ab_{tag}_{i}.log').read()
d = json.loads(txt.strip().splitlines()[-1])
tot = dict(re.findall(r'"(vqa_[a-z0-9_]+\[[a-z_]+\])": ([0-9.]+)', txt))
keys = ['vqa_conv3x3_fwd[fwd]', 'vqa_conv3x3_fwd[dgrad]', 'vqa_conv3x3_wgrad[wgrad]', 'vqa_conv3x3_first_fwd[first]']
print(tag, i, d['value'], d['ms_per_step'], ' '.join(f"{k.split('[')[1][:-1]}={tot.get(k)}" for k in keys))
PY
  done
done
