"""Per-kernel average durations of two rocprofv3 --stats runs side by side: python scripts/prof_compare.py A_stats.csv B_stats.csv"""
import csv, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r['Name']] = (int(r['Calls']), float(r['TotalDurationNs']))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
rows.sort(key=lambda r: -abs(r[0]))
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
print(f'total A {sum(v[1] for v in a.values()) / steps / 1e6:.3f} ms/step   B {sum(v[1] for v in b.values()) / steps / 1e6:.3f} ms/step')
for d, k, ca, ta, cb, tb in rows[:28]:
    print(f'{d / steps / 1e3:+8.1f} us/step  A {ca:5d} x {ta / max(ca, 1) / 1e3:8.1f} us   B {cb:5d} x {tb / max(cb, 1) / 1e3:8.1f} us   {k[:90]}')
