"""Group a rocprofv3 kernel trace by (hardware queue id, HIP stream id): python3 scripts/dp_queues_summarise.py DIR OUT.json
Kernels of the timed steps only (after the first launch of the trainer's stream = the stream with the most kernels)."""
import collections
import csv
import glob
import json
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
per = collections.Counter((r['Queue_Id'], r['Stream_Id']) for r in rows)
main = per.most_common(1)[0][0]
t_first = min(int(r['Start_Timestamp']) for r in rows if (r['Queue_Id'], r['Stream_Id']) == main)
groups = collections.defaultdict(collections.Counter)
before = collections.Counter()
for r in rows:
    key = (r['Queue_Id'], r['Stream_Id'])
    if int(r['Start_Timestamp']) < t_first:
        before[key] += 1
        continue
    name = r['Kernel_Name']
    name = name[name.find('::') + 2:] if '::' in name else name
    groups[key][name.split('<')[0].split('(')[0][:48]] += 1
out = dict(trainer_stream=dict(queue_id=main[0], stream_id=main[1]), during_steps=[], before_first_step=[])
for (q, s), c in sorted(groups.items(), key=lambda kv: -sum(kv[1].values())):
    out['during_steps'].append(dict(queue_id=q, stream_id=s, kernels=sum(c.values()),
                                    top=[f'{k} x{n}' for k, n in c.most_common(5)]))
    print('steps  queue', q, 'stream', s, sum(c.values()), c.most_common(3))
for (q, s), n in sorted(before.items(), key=lambda kv: -kv[1]):
    out['before_first_step'].append(dict(queue_id=q, stream_id=s, kernels=n))
    print('before queue', q, 'stream', s, n)
out['distinct_queues_during_steps'] = sorted({g['queue_id'] for g in out['during_steps']})
json.dump(out, open(sys.argv[2], 'w'), indent=1)
