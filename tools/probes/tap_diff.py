"""Compare two tap files of post_addend_corruption_probe.py --taps."""
import sys
import torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in a:
    if k not in b:
        print("missing in b:", k)
        continue
    x, y = a[k].float(), b[k].float()
    if x.shape != y.shape:
        print(f"{k}: shape {tuple(x.shape)} vs {tuple(y.shape)}")
        continue
    fin = torch.isfinite(x) & torch.isfinite(y)
    nbad = int((~torch.isfinite(y)).sum()) - int((~torch.isfinite(x)).sum())
    rel = ((x - y)[fin].norm() / (x[fin].norm() + 1e-20)).item() if fin.any() else float("nan")
    flag = "  <<<<" if (rel > 2e-2 or nbad != 0) else ""
    print(f"{k:34s} {str(tuple(x.shape)):18s} rel {rel:.3e} |a| {x[fin].norm().item():.3e} |b| {y[fin].norm().item():.3e} extra-nonfinite {nbad}{flag}")
