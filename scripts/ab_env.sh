#!/bin/bash
# Same-box A/B of run-time variants: scripts/ab_env.sh ROUNDS "name|ENV=.. ENV2=.." ...
# alternates `bench.py --steps 60 --warmup 15` runs of every variant for ROUNDS rounds and prints ms/step of each
# (variants are compared inside ONE gpurun call: box-to-box spread is 2-3 %, run-to-run on one box ~1-2 %).
rounds=$1; shift
mkdir -p gpurun_out/ab
for i in $(seq $rounds); do
  for spec in "$@"; do
    name=${spec%%|*}; envs=${spec#*|}
    env $envs timeout -k 10 150 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --breakdown > gpurun_out/ab/${name}_$i.log 2>&1 || { echo "$name FAILED"; tail -5 gpurun_out/ab/${name}_$i.log; exit 1; }
    python - "$name" "$i" <<'PY'
import json, sys
name, i = sys.argv[1], sys.argv[2]
d = json.loads(open(f'gpurun_out/ab/{name}_{i}.log').read().strip().splitlines()[-1])
print(f"{name:14s} round {i}: {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} samples/s  kernels {d.get('device_ms_in_kernels')}", flush=True)
PY
  done
done
