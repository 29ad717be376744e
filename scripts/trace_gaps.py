import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# take the last full step: find last occurrence of nchw_to_nhwc start
idx = [i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name']]
s, e = idx[-2], idx[-1]
step = rows[s:e]
t0 = int(step[0]['Start_Timestamp'])
busy_end = t0
gap_total = 0
out = []
for r in step:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = st - busy_end
    if gap > 0: gap_total += gap
    out.append(((st - t0) / 1e3, (en - st) / 1e3, max(gap, 0) / 1e3, r['Kernel_Name'][:60]))
    busy_end = max(busy_end, en)
print('step span us', (busy_end - t0) / 1e3, 'idle us', gap_total / 1e3, 'kernels', len(step))
big = sorted(out, key=lambda x: -x[2])[:25]
for o in sorted(big): print('t=%9.1f dur=%8.1f gap_before=%7.1f %s' % o)
