"""Is a hipMemsetAsync captured into a HIP graph ordered against the kernels around it?  gps_gather_points_grad =
memset(grad_points) + an accumulate kernel; captured and replayed, compared with the eager result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sceneverse_amd.pointnet2 import _ext as hip
dev = torch.device("cuda", 0)
torch.manual_seed(0)
b, c, n, m = 64, 128, 1024, 4096
go = torch.randn(b, c, m, device=dev)
idx = torch.randint(0, n, (b, m), device=dev, dtype=torch.int32)
ref = hip.gather_points_grad(go, idx, n).clone()
s = torch.cuda.Stream()
bad = 0
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        outs = [hip.gather_points_grad(go, idx, n) for _ in range(4)]
        filler = [torch.full((1 << 22,), float(i), device=dev) for i in range(4)]   # other work between / after
    for it in range(50):
        g.replay(); torch.cuda.synchronize()
        for o in outs:
            d = (o - ref).abs().max().item()
            if not d <= 1e-3:
                bad += 1
                print("replay", it, "max diff", d, flush=True)
print("bad", bad, "of", 200)
