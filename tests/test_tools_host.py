"""Host-side measurement tools (no GPU): tools/step_from_trace.py picks ONE steady-state step out of a kernel trace --
the middle window of the longest run of inter-marker windows with the same dispatch count -- and splits it into
libgps_hip.so kernels and everything else; tools/update_step_bounds.py turns the [bf16-vs-fp32] lines of a test log
into the golden file's format."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path):
    rows, t = [], 0

    def add(name, dur):
        nonlocal t
        rows.append({"Kind": "KERNEL_DISPATCH", "Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + dur})
        t += dur + 10
    for _ in range(50):                                   # start-up work: fills and copies that a --stats table would count
        add("void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float> >", 3000)
    for step in range(7):
        extra = 5 if step in (0, 6) else 0                # a warm-up step and a trailing eager timing pass launch more
        add("gps_gemm::wgrad_grouped_kernel(int, int, int, int)", 1_000_000)
        for _ in range(4):
            add("void gps_gemm::gemm_kernel<128, 128, 4, 2, false, false, 0, 2, false, false>(gps_gemm::Params)", 40_000)
        for _ in range(3 + extra):
            add("void at::native::vectorized_elementwise_kernel<4, at::native::CUDAFunctor_add<float> >", 7_000)
        add("__amd_rocclr_copyBuffer", 4_000)
        add("Cijk_Alik_Bljk_S_B_Bias_HA_S_SAV_UserArgs_MT96x160x16", 50_000)
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


def test_step_from_trace_picks_a_steady_state_window(tmp_path):
    trace, out = str(tmp_path / "trace.csv"), str(tmp_path / "step.json")
    _trace(trace)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_from_trace.py"), trace, "--json", out],
                       capture_output=True, text=True, check=True)
    assert "-> window 3" in r.stdout, r.stdout           # windows 1..4 have 10 dispatches each: the middle of that run
    d = json.load(open(out))
    assert d["launches"] == 10
    assert d["libgps_hip"] == {"us": 1160.0, "launches": 5}
    assert d["outside"]["launches"] == 5 and abs(d["outside"]["us"] - 75.0) < 1e-6
    fam = d["outside"]["by_family"]
    assert fam["torch"] == {"us": 21.0, "launches": 3} and fam["hipBLASLt"]["launches"] == 1 and fam["rocclr copy / fill"]["launches"] == 1


def test_update_step_bounds_parses_the_test_lines(tmp_path):
    log, out = tmp_path / "log.txt", tmp_path / "measured.json"
    log.write_text("noise\n....[bf16-vs-fp32] loss total_loss: 0.0003\n[bf16-vs-fp32] rel bert encoder.layer.0.attention.self.query.weight: 0.0376\n")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "update_step_bounds.py"), str(log), "unit test", str(out)], check=True,
                   capture_output=True)
    got = json.load(open(out))
    assert got["source"] == "unit test"
    assert got["measured"] == {"loss total_loss": 0.0003, "rel bert encoder.layer.0.attention.self.query.weight": 0.0376}
    # the committed golden file has this format (tests/test_gpu_model.py reads it)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "bf16_step_measured.json")))
    assert set(golden) == {"source", "measured"} and all(isinstance(v, float) for v in golden["measured"].values())
