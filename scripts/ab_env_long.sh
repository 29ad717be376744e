#!/bin/bash
# scripts/ab_env_long.sh ROUNDS "name|ENV=.." ...: like ab_env.sh with 150 timed steps after 30 of warm-up and no roofline
# breakdown - for effects of a few tenths of a per cent
rounds=$1; shift
mkdir -p gpurun_out/abl
for i in $(seq $rounds); do
  for spec in "$@"; do
    name=${spec%%|*}; envs=${spec#*|}
    env $envs timeout -k 10 150 python bench.py --steps 150 --warmup 30 --no-cpu-baseline > gpurun_out/abl/${name}_$i.log 2>&1 || { echo "$name FAILED"; tail -5 gpurun_out/abl/${name}_$i.log; exit 1; }
    python -c "import json,sys; d=json.loads(open('gpurun_out/abl/${name}_$i.log').read().strip().splitlines()[-1]); print('%-12s round $i: %.3f ms/step' % ('$name', d['ms_per_step']), flush=True)"
  done
done
