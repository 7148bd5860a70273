"""Diagnostics only (tools/probes): named taps that keep a COPY of a tensor at a point of the step.  Inside a HIP-graph
capture the copy is a captured kernel into a buffer this module keeps alive, so after a replay `TAPS[name]` holds what
the tensor held at that point of the replay.  Off unless a probe switches it on; the product path never reads it."""
import torch

ENABLED = False
TAPS = {}
_COUNT = {}


def tap(name: str, t) -> None:
    if not ENABLED or t is None:
        return
    k = _COUNT.get(name, 0)
    _COUNT[name] = k + 1
    TAPS[f"{name}#{k}"] = t.detach().clone()


def reset() -> None:
    TAPS.clear()
    _COUNT.clear()
