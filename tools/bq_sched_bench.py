"""ball_query block-schedule tuning (gps::BqSched, GPS_BQ_SCHED=0..6 is read once per process):
    for s in 0 1 2 3 4 5 6; do GPS_BQ_SCHED=$s python tools/bq_sched_bench.py; done
Times SA1 ball_query (r = 0.2, 32 samples) on the bench batch (64 scenes x 80 objects x 1024 points, padding objects
included) and on the stress cloud size (2048 points), as a 10-call HIP-graph replay; every schedule must give the bits
of schedule 0 (the first run leaves its indices in /tmp)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sceneverse_amd.data.synthetic import synth_batch
from sceneverse_amd.pointnet2 import _ext as hip

dev = "cuda"
sched = os.environ.get("GPS_BQ_SCHED", "default")
out = {"sched": sched}
for tag, n_pts, batch in (("1024", 1024, 64), ("2048", 2048, 8)):
    b = synth_batch(batch, n_obj=80, n_pts=n_pts, seed=42, device=dev)
    xyz = b["obj_fts"].reshape(-1, n_pts, 6)[..., :3].contiguous()
    fps = hip.furthest_point_sampling(xyz, 32)
    new_xyz = hip.gather_points(xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    idx = hip.ball_query(new_xyz, xyz, 0.2, 32)
    ref_path = f"/tmp/bq_ref_{tag}.pt"
    if os.path.exists(ref_path):
        assert torch.equal(idx.cpu(), torch.load(ref_path)), f"schedule {sched}: indices differ from schedule 0 ({tag})"
    else:
        torch.save(idx.cpu(), ref_path)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            hip.ball_query(new_xyz, xyz, 0.2, 32)
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 100.0)
    out[f"us_{tag}"] = round(best, 2)
print(out)
