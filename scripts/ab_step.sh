#!/bin/bash
# Same-box A/B of the step time alone: scripts/ab_step.sh <base.so> <new.so> [rounds] [steps]
base=$1; new=$2; rounds=${3:-4}; steps=${4:-60}
for i in $(seq $rounds); do
  for tag in base new; do
    lib=$base; [ $tag = new ] && lib=$new
    VQA_HIP_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps $steps --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', $i, d['ms_per_step'], d['roofline_by_kernel']['conv3x3_igemm_p']['kernel_ms_per_step'], d['roofline_by_kernel']['conv3x3_igemm_tile']['kernel_ms_per_step'])" || exit 1
  done
done
