"""tests/golden/bf16_step_measured.json from the [bf16-vs-fp32] lines of
`pytest -s tests/test_gpu_model.py -k fp32_oracle_port` (one MI355X run):   python tools/update_step_bounds.py LOG "source text" [OUT.json] """
import json, os, re, sys
log, source = sys.argv[1], sys.argv[2]
meas = {}
for line in open(log):
    m = re.search(r"\[bf16-vs-fp32\] (.+): ([0-9.eE+-]+)\s*$", line)
    if m:
        meas[m.group(1)] = float(m.group(2))
path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                                          "bf16_step_measured.json")
json.dump({"source": source, "measured": meas}, open(path, "w"), indent=1)
print(len(meas), "quantities ->", path)
