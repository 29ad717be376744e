#!/bin/bash
# Same-box A/B of the data-parallel step on one rank (bench.py --force-buckets: hooks, bucketed RCCL all-reduces, joins)
# against the plain step, and of its two stream choices.  usage: scripts/dp_ab.sh OUTDIR [steps]
set -u
out=${1:-gpurun_out/dp_ab}; steps=${2:-100}
mkdir -p "$out"
run() {  # name, env assignments..., then -- bench args
    name=$1; shift
    envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout -k 10 300 python bench.py --steps "$steps" --warmup 20 --no-cpu-baseline "$@" \
        > "$out/$name.json" 2> "$out/$name.err" || { echo "$name failed"; tail -5 "$out/$name.err"; return 1; }
    python - "$out/$name.json" "$name" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(f"{sys.argv[2]:28s} {r['ms_per_step']:.3f} ms/step  {r['value']:.0f} samples/s")
PY
}
for rep in 1 2; do
    run plain_$rep X=1 -- || exit 1
    run fb_$rep X=1 -- --force-buckets || exit 1
    run fb_fold_$rep VQA_DP_FOLD_STREAMS=1 -- --force-buckets || exit 1
    run fb_hookstream_$rep VQA_DP_ISSUE_FROM_PRODUCER=0 -- --force-buckets || exit 1
done
