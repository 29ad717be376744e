import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in a:
    x, y = a[k].double(), b[k].double()
    d = float((x - y).norm() / (y.norm() + 1e-30))
    if k.startswith('loss') or k.startswith('grad') or d > 1e-5:
        print(f'{k:50s} rel diff {d:.3e}  norm {float(y.norm()):.5g}')
