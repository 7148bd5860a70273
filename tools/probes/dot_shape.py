"""Is a dumped HIP graph a linear chain?  Counts nodes, edges, fan-out / fan-in > 1."""
import collections
import re
import sys
txt = open(sys.argv[1]).read()
edges = re.findall(r'"?([\w\.]+)"?\s*->\s*"?([\w\.]+)"?', txt)
out, inn = collections.Counter(a for a, _ in edges), collections.Counter(b for _, b in edges)
nodes = set(out) | set(inn)
print(f"{sys.argv[1]}: {len(nodes)} nodes, {len(edges)} edges, fan-out>1: {sum(1 for v in out.values() if v > 1)}, "
      f"fan-in>1: {sum(1 for v in inn.values() if v > 1)}, roots: {len(nodes - set(inn))}, sinks: {len(nodes - set(out))}")
for n, v in out.most_common(5):
    if v > 1:
        print("  fan-out", n, v)
