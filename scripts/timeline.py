"""Timeline of one steady-state training step from a rocprofv3 kernel trace of the overlapped bench:
python3 scripts/timeline.py DIR [step index from the end, default 3]
Per stream: busy time and idle gaps; for the trainer's stream (the one that runs conv3x3_first_fwd) every kernel with
its start offset, duration and the gap in front of it - where the critical chain waits."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))


def short(n):
    n = n[n.find('::') + 2:] if '::' in n else n
    return n.split('(')[0][:70]


firsts = [i for i, r in enumerate(rows) if 'conv3x3_first_fwd' in r['Kernel_Name']]
s, e = firsts[-back - 1], firsts[-back]
main = (rows[s]['Queue_Id'], rows[s]['Stream_Id'])
t0, t1 = int(rows[s]['Start_Timestamp']), int(rows[e]['Start_Timestamp'])
step = [r for r in rows if t0 <= int(r['Start_Timestamp']) < t1]
print(f'step span {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels')
per = collections.defaultdict(list)
for r in step:
    per[(r['Queue_Id'], r['Stream_Id'])].append(r)
for key, rs in per.items():
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    print(f'stream {key}{" (trainer)" if key == main else ""}: {len(rs)} kernels, busy {busy / 1e3:.1f} us')
prev = t0
gaps = 0
print('--- trainer stream')
for r in per[main]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = st - prev
    gaps += max(gap, 0)
    # side-stream kernels running during this one
    ov = [short(o['Kernel_Name'])[:28] for k, rs in per.items() if k != main for o in rs
          if int(o['Start_Timestamp']) < en and int(o['End_Timestamp']) > st]
    print(f't={(st - t0) / 1e3:8.1f} dur={(en - st) / 1e3:7.1f} gap={gap / 1e3:6.1f}  {short(r["Kernel_Name"])[:60]:60s} | {len(ov)} side: {", ".join(sorted(set(ov)))[:80]}')
    prev = en
print(f'trainer stream idle inside the step: {gaps / 1e3:.1f} us (+ {(t1 - prev) / 1e3:.1f} us before the next step)')
