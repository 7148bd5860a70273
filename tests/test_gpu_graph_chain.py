"""Root cause of round 3's "post-addend" corruption, as a regression test.

A parameter's AccumulateGrad node remembers the stream that was current when it was created; it outlives a step when
something still references that step's autograd graph (round 3: `model._stage_boundary`), and round 3 ran every warm-up
step on a fresh side stream.  Inside the capture of the bottom backward graph the stale nodes issued `grad += dW` on
THEIR stream: the captured graph forked (the accumulations became parallel branches), ROCm's multi-queue graph executor
ran nodes of such a graph out of order, and later kernels of the same graph overwrote activations the text encoder had
saved (wrong gradients; a memory-aperture violation when the encoders ran in the other order).

The test captures the step's graphs in a child process (tests/helpers/graph_chain_child.py) and reads their topology from
the runtime (hipGraphGetNodes / hipGraphGetEdges on the kept hipGraph_t): every graph of the CURRENT engine must be a
linear chain (no node with two successors); a subclass with round 3's stream handling (`--legacy`) must show the fork --
the failure mode being guarded against.  Nothing here skips: a ROCm build on which the topology cannot be read fails."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph_shapes(tmp_path, *flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "graph_chain_child.py"), *flags],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("captured ")]
    assert lines, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    shapes = json.loads(lines[-1][len("captured "):])
    assert shapes and all(s["nodes"] > 0 for s in shapes), shapes
    return shapes


@pytest.mark.timeout(900)
@pytest.mark.parametrize("flags", [(), ("--classic-wgrad",)])
def test_captured_graphs_of_the_split_graph_step_are_linear_chains(tmp_path, flags):
    shapes = _graph_shapes(tmp_path, *flags)
    big = [s for s in shapes if s["edges"] >= 20]
    assert len(big) >= 3, shapes                                # forward | losses + top backward | bottom backward(s)
    assert all(s["forks"] == 0 for s in shapes), shapes
    assert all(s["edges"] == s["nodes"] - 1 for s in shapes), shapes      # a chain: one edge less than nodes


@pytest.mark.timeout(900)
def test_round3_stream_handling_forks_the_bottom_backward_graph(tmp_path):
    shapes = _graph_shapes(tmp_path, "--legacy", "--classic-wgrad")
    assert any(s["forks"] > 0 for s in shapes), shapes          # the reproducer: stale AccumulateGrad nodes fork the capture
