"""Builds libgps_hip.so (gfx950) in-tree with hipcc.  No torch headers are needed: the library
is a plain C-ABI shared object (include/gps_hip.h)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libgps_hip.so")

# (source, extra flags).  The point ops pin fp32 rounding (-ffp-contract=off);
# the transformer kernels want FMA contraction and are compiled separately.
SOURCES = [
    ("gps_point_ops.hip", ["-ffp-contract=off"]),
    ("gps_sa_mlp.hip", []),
    ("gps_attention.hip", []),
    ("gps_losses.hip", []),
    ("gps_layernorm.hip", []),
    ("gps_objects.hip", []),
    ("gps_reduce.hip", []),
    ("gps_embedding.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
          "-I" + os.path.join(ROOT, "include")]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libgps_hip.so cannot be built on this machine")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s, _ in SOURCES if os.path.exists(os.path.join(HERE, s))]
    deps += [os.path.join(ROOT, "include", "gps_hip.h"), os.path.abspath(__file__)]
    deps += [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".h", ".hpp"))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cc = hipcc()
    objs = []
    for src, extra in SOURCES:
        path = os.path.join(HERE, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(HERE, src.rsplit(".", 1)[0] + ".o")
        cmd = [cc, *COMMON, *extra, "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
