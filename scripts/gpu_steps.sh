#!/bin/bash
# Run GPU steps one after the other, each under its own time limit; stop at the first step that ends in anything but
# "passed" (0) or "some tests failed" (1): after a fault, abort, kill or timeout nothing more is started on the GPU.
#   scripts/gpu_steps.sh LOGDIR "limit_seconds|name|command" ...
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
    lim=${spec%%|*}; rest=${spec#*|}; name=${rest%%|*}; cmd=${rest#*|}
    echo "=== $name (limit ${lim}s): $cmd"
    timeout -k 10 "$lim" bash -c "$cmd" > "$out/$name.log" 2>&1
    rc=$?
    echo "=== $name rc=$rc"; tail -n 25 "$out/$name.log"
    if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: $name ended with $rc"; exit $rc; fi
done
