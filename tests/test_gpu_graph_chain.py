"""Root cause of round 3's "post-addend" corruption, as a regression test.

A parameter's AccumulateGrad node remembers the stream that was current when it was created; it outlives a step when
something still references that step's autograd graph (round 3: `model._stage_boundary`), and round 3 ran every warm-up
step on a fresh side stream.  Inside the capture of the bottom backward graph the stale nodes issued `grad += dW` on
THEIR stream: the captured graph forked (the accumulations became parallel branches), ROCm's multi-queue graph executor
ran nodes of such a graph out of order, and later kernels of the same graph overwrote activations the text encoder had
saved (wrong gradients; a memory-aperture violation when the encoders ran in the other order).

The test instantiates the step's graphs in a child process with DEBUG_HIP_GRAPH_DOT_PRINT=1 and parses the DOT files ROCm
writes: every graph of the CURRENT engine must be a linear chain (no node with two successors); the same engine with round
3's stream handling (`--legacy`) must show the fork -- that is the failure mode being guarded against."""
import collections
import glob
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph_shapes(tmp_path, *flags):
    env = dict(os.environ, DEBUG_HIP_GRAPH_DOT_PRINT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probes", "graph_chain_check.py"), *flags],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert "captured" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    shapes = []
    for f in sorted(glob.glob(os.path.join(str(tmp_path), "graph_*"))):
        edges = re.findall(r'"([\w\.]+)"\s*->\s*"([\w\.]+)"', open(f).read())
        out = collections.Counter(a for a, _ in edges)
        shapes.append((os.path.basename(f), len(edges), sum(1 for v in out.values() if v > 1)))
    return shapes


@pytest.mark.timeout(900)
@pytest.mark.parametrize("flags", [(), ("--classic-wgrad",)])
def test_captured_graphs_of_the_split_graph_step_are_linear_chains(tmp_path, flags):
    shapes = _graph_shapes(tmp_path, *flags)
    big = [s for s in shapes if s[1] >= 20]
    if not shapes:
        pytest.skip("this ROCm build does not write DOT files for DEBUG_HIP_GRAPH_DOT_PRINT")
    assert len(big) >= 3, shapes                                # forward | losses + top backward | bottom backward
    assert all(forks == 0 for _, _, forks in shapes), shapes


@pytest.mark.timeout(900)
def test_round3_stream_handling_forks_the_bottom_backward_graph(tmp_path):
    shapes = _graph_shapes(tmp_path, "--legacy", "--classic-wgrad")
    if not shapes:
        pytest.skip("this ROCm build does not write DOT files for DEBUG_HIP_GRAPH_DOT_PRINT")
    assert any(forks > 0 for _, _, forks in shapes), shapes     # the reproducer: stale AccumulateGrad nodes fork the capture
