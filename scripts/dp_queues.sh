#!/bin/bash
# Which hardware queue does each live stream of the data-parallel step land on?  rocprofv3 kernel trace of the 1-rank
# RCCL step (bench.py --force-buckets), kernels grouped by (Queue_Id, Stream_Id).  Writes gpurun_out/dp_queues.json.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/dp_queues
rm -rf $OUT; mkdir -p $OUT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29747 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --force-buckets "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob, json, collections
f = glob.glob('$OUT/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
cols = rows[0].keys()
skey = 'Stream_Id' if 'Stream_Id' in cols else None
groups = collections.defaultdict(collections.Counter)
for r in rows:
    groups[(r['Queue_Id'], r[skey] if skey else '?')][r['Kernel_Name'].split('(')[0][-60:]] += 1
out = []
for (q, s), c in sorted(groups.items(), key=lambda kv: -sum(kv[1].values())):
    out.append(dict(queue_id=q, stream_id=s, kernels=sum(c.values()), top=[f'{k} x{n}' for k, n in c.most_common(4)]))
    print(q, s, sum(c.values()), c.most_common(3))
json.dump(dict(columns=list(cols), groups=out), open('gpurun_out/dp_queues.json', 'w'), indent=1)
PY
tail -1 $OUT/bench.log | cut -c1-200
