import sys, math, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_attention as T
from sceneverse_amd import _native
from sceneverse_amd.modules.layers.fused_attention import _FusedSelfAttention
lib = _native.load()
def rel(a, b): return ((a.float().cpu() - b).norm() / b.norm()).item()
for (B, L, spatial) in [(8, 130, False), (8, 80, True), (8, 50, False)]:
    packed, pl, mask = T._inputs(B, L, spatial, seed=7)
    ref_in = packed.float().requires_grad_(True)
    ref = T.ref_attention(ref_in, pl, mask)
    go = torch.randn(B, L, T.D, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16)
    ref.backward(go.float())
    for fam, thr in (("resident", (10, 10)), ("streaming", (1, 1))):
        lib.gps_attn_set_stream_min_tiles(*thr)
        x = packed.to("cuda").requires_grad_(True)
        out = _FusedSelfAttention.apply(x, pl.to("cuda") if pl is not None else None, mask.to("cuda") if mask is not None else None, T.H, 0.0, 0, None)
        out.backward(go.to("cuda"))
        g = x.grad.float().cpu()
        D = T.D
        print(B, L, spatial, fam, "out %.2e dq %.2e dk %.2e dv %.2e" % (rel(out, ref.detach()), rel(g[..., :D], ref_in.grad[..., :D]), rel(g[..., D:2*D], ref_in.grad[..., D:2*D]), rel(g[..., 2*D:3*D], ref_in.grad[..., 2*D:3*D])))
