"""The north star's ball_query + group figure alone (bench.bq_group_unfused on the bench preset's batch), repeated.

    python tools/bq_headline.py [--repeat 3]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--config", default="pretrain")
    args = ap.parse_args()
    import bench
    from sceneverse_amd.data.synthetic import synth_batch
    preset = bench.WORKLOADS[args.config]
    dev = torch.device("cuda", 0)
    batch = synth_batch(preset["batch"], n_obj=preset["n_obj"], n_pts=preset["n_pts"], txt_len=preset["txt_len"],
                        seed=42, device=dev)
    for _ in range(args.repeat):
        r = bench.bq_group_unfused(batch)
        print(json.dumps({"frac_hbm": r["frac_hbm"], "us": r["us"],
                          "launches": {x["op"]: x["us"] for x in r["launches"]}}))


if __name__ == "__main__":
    main()
