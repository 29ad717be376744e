#!/bin/bash
# Which hardware queue does each live stream of the data-parallel step land on?  rocprofv3 kernel trace of the 1-rank
# RCCL step (bench.py --force-buckets), kernels grouped by (Queue_Id, Stream_Id).  Writes gpurun_out/dp_queues.json.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/dp_queues
rm -rf $OUT; mkdir -p $OUT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29747 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --force-buckets "$@" > $OUT/bench.log 2>&1
python3 scripts/dp_queues_summarise.py $OUT gpurun_out/dp_queues.json
tail -1 $OUT/bench.log | cut -c1-200
