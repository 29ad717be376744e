"""Parse a rocprofv3 kernel trace of scripts/gemm_micro.py: true device time per (shape, splits) config."""
import csv, glob, sys
sys.path.insert(0, 'scripts')
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
calls = []   # device time per vqa_gemm call (gemm kernel + its split-K reduce)
for r in rows:
    n = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'gemm_bf16' in n or 'gemm_kernel' in n: calls.append(d)
    elif 'gemm_splitk_reduce' in n: calls[-1] += d
import re
shapes = re.findall(r"\((\d), (\d+), (\d+), (\d+), '([^']+)'\)", open('scripts/gemm_micro.py').read())
per = 12
for i, (t, M, N, K, label) in enumerate(shapes):
    res = []
    for j in range(6):
        c = calls[(i * 6 + j) * per:(i * 6 + j + 1) * per]
        res.append(min(c[4:]))
    fl = 2e-9 * int(M) * int(N) * int(K)
    print('%-14s %s %5sx%5sx%5s auto %6.1f us %6.1f TF/s | 1,2,4,8,16: %s' % (label, 'TN' if t == '1' else 'NT', M, N, K, res[0], fl / res[0] * 1e3, ' '.join('%6.1f' % x for x in res[1:])))
