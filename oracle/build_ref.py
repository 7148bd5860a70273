"""Build oracle/_ref/pointnet2_ref_ext*.so: the REFERENCE's own pointnet2 extension
(/root/reference/modules/third_party/pointnet2/_ext_src, 4 .cu + 5 .cpp files), compiled
unmodified for gfx950 through torch.utils.cpp_extension (which hipifies CUDA sources on ROCm).

TEST INFRASTRUCTURE ONLY.  The sources are compiled from where they lie: they are staged in a
scratch directory under /tmp for the hipify pass (the reference tree is read-only) and are never
copied into this repository; only the resulting shared object lands in oracle/_ref/ (git-ignored,
but shipped to the GPU box by gpurun).  The extension is GPU-only -- every host wrapper asserts
"CPU not supported" -- so it is exercised by the `-m gpu` tests, where it is the second,
independent oracle for the nine native ops ("the reference itself, run here").

The reference's own setup.py is NOT used (its nvcc-only flags `-Xfatbin -compress-all`,
setup.py:32, break on ROCm).
"""
from __future__ import annotations

import glob
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
REF_EXT = "/root/reference/modules/third_party/pointnet2/_ext_src"
NAME = "pointnet2_ref_ext"


def built_path() -> str | None:
    hits = sorted(glob.glob(os.path.join(OUT_DIR, NAME + "*.so")))
    return hits[0] if hits else None


def build(force: bool = False, verbose: bool = False) -> str | None:
    """Returns the .so path, or None when /root/reference is absent (e.g. on the GPU box, where
    the prebuilt file is used)."""
    have = built_path()
    if have and not force:
        return have
    if not os.path.isdir(REF_EXT):
        return have
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    from torch.utils.cpp_extension import load
    stage = tempfile.mkdtemp(prefix="gps_ref_ext_")
    try:
        shutil.copytree(REF_EXT, os.path.join(stage, "_ext_src"))
        src = sorted(glob.glob(os.path.join(stage, "_ext_src", "src", "*.cpp")) +
                     glob.glob(os.path.join(stage, "_ext_src", "src", "*.cu")))
        build_dir = os.path.join(stage, "build")
        os.makedirs(build_dir)
        load(name=NAME, sources=src, extra_include_paths=[os.path.join(stage, "_ext_src", "include")],
             extra_cflags=["-O2"], extra_cuda_cflags=["-O2"], build_directory=build_dir,
             verbose=verbose, is_python_module=False)
        os.makedirs(OUT_DIR, exist_ok=True)
        so = glob.glob(os.path.join(build_dir, NAME + "*.so"))[0]
        dst = os.path.join(OUT_DIR, os.path.basename(so))
        shutil.copy2(so, dst)
        return dst
    finally:
        shutil.rmtree(stage, ignore_errors=True)


def load_ext():
    """Import the prebuilt reference extension as a python module exposing the 9 `_ext`
    functions (GPU tensors only)."""
    path = built_path()
    if path is None:
        raise FileNotFoundError("oracle/_ref/pointnet2_ref_ext*.so not built (run oracle/build_ref.py "
                                "where /root/reference exists)")
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
