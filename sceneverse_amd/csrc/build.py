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
    ("gps_attention_ex.hip", []),
    ("gps_attention_sp.hip", []),
    ("gps_attention_fa.hip", []),
    ("gps_losses.hip", []),
    ("gps_contrastive.hip", []),
    ("gps_layernorm.hip", []),
    ("gps_objects.hip", []),
    ("gps_reduce.hip", []),
    ("gps_embedding.hip", []),
    ("gps_bert_embed.hip", []),
    ("gps_loc_embed.hip", []),
    ("gps_gemm.hip", []),
    ("gps_optim.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
          "-I" + os.path.join(ROOT, "include")]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libgps_hip.so cannot be built on this machine")
    return exe


def hipcc_available() -> bool:
    return bool(shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"))


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s, _ in SOURCES if os.path.exists(os.path.join(HERE, s))]
    deps += [os.path.join(ROOT, "include", "gps_hip.h"), os.path.abspath(__file__)]
    deps += [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".h", ".hpp"))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile + link under an exclusive file lock and publish the library with an atomic rename: when the N
    ranks of a torchrun job reach their first native call together, one of them builds, the others wait on the
    lock and then find an up-to-date library (never a half-written one)."""
    if not force and not needs_build():
        return LIB
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():        # another process built it while we waited
                return LIB
            cc = hipcc()
            jobs = []
            for src, extra in SOURCES:
                path = os.path.join(HERE, src)
                if not os.path.exists(path):
                    continue
                obj = os.path.join(HERE, src.rsplit(".", 1)[0] + ".o")
                jobs.append(([cc, *COMMON, *extra, "-c", path, "-o", obj], obj))

            def run(job):
                if verbose:
                    print(" ".join(job[0]))
                subprocess.check_call(job[0])
                return job[1]

            with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
                objs = list(pool.map(run, jobs))
            tmp = f"{LIB}.tmp.{os.getpid()}"
            cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
