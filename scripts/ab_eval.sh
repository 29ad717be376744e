#!/bin/bash
# scripts/ab_eval.sh ROUNDS "name|ENV=.." ...: BASELINE.json configs[3] (inference, batch 1024), 60 timed steps after 15
rounds=$1; shift
mkdir -p gpurun_out/abe
for i in $(seq $rounds); do
  for spec in "$@"; do
    name=${spec%%|*}; envs=${spec#*|}
    env $envs timeout -k 10 150 python bench.py --mode eval --batch 1024 --steps 60 --warmup 15 > gpurun_out/abe/${name}_$i.log 2>&1 || { echo "$name FAILED"; tail -5 gpurun_out/abe/${name}_$i.log; exit 1; }
    python -c "import json,sys; d=json.loads(open('gpurun_out/abe/${name}_$i.log').read().strip().splitlines()[-1]); print('%-12s round $i: %.3f ms per 1024 = %.0f samples/s' % ('$name', d['ms_per_step'], d['value']), flush=True)"
  done
done
