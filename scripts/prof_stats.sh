#!/bin/bash
# rocprofv3 kernel statistics of a bench.py run: scripts/prof_stats.sh OUTDIR [bench.py args...]
# (the program itself follows `--`: no env / shell wrapper between the profiler and python3)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- python3 bench.py "$@" > "$out/bench.log" 2>&1
rc=$?
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
echo "rc=$rc stats=$f"; tail -1 "$out/bench.log" | cut -c1-200
[ -n "$f" ] && head -40 "$f" | cut -c1-160
exit $rc
