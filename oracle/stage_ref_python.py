"""Build oracle/_ref/ref_python.zip: the reference's own trainer-side Python packages, byte for byte, as ONE archive
that travels to the GPU box next to oracle/_ref/pointnet2_ref_ext.so.

TEST INFRASTRUCTURE ONLY.  /root/reference does not exist on the GPU box, so the `-m gpu` variant of
tests/test_reference_trainer_dropin.py (the reference's unmodified `DefaultTrainer` driving OUR model classes on
libgps_hip.so) had never executed anywhere.  This script packs the packages that test imports

    trainer/  optim/  evaluator/  common/  data/          (python files only)
    modules/  model/                                       (the reference's own model: bench.py's CPU baseline of kind
                                                            "reference", oracle/ref_cpu_baseline.py -- SURVEY.md 8(d))

from where they lie into a zip the test puts on sys.path (zipimport) when /root/reference is absent.  Nothing is
unpacked into the repository, the archive is git-ignored (oracle/_ref/), and no product module ever opens it.
`__graft_entry__.build()` runs this where /root/reference exists; the GPU box only uses the prebuilt file.
"""
from __future__ import annotations

import os
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref", "ref_python.zip")
REF = "/root/reference"
PACKAGES = ("trainer", "optim", "evaluator", "common", "data", "modules", "model")


def built_path() -> str | None:
    return OUT if os.path.exists(OUT) else None


def build(force: bool = False) -> str | None:
    """-> path of the archive, or None when neither the archive nor /root/reference exists."""
    if os.path.exists(OUT) and not force:
        return OUT
    if not os.path.isdir(os.path.join(REF, "trainer")):
        return built_path()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    tmp = OUT + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for pkg in PACKAGES:
            for d, _, files in sorted(os.walk(os.path.join(REF, pkg))):
                if "__pycache__" in d:
                    continue
                # explicit directory entries: zipimport resolves a package without __init__.py (`common`) as a
                # namespace package only when its directory is listed
                z.writestr(zipfile.ZipInfo(os.path.relpath(d, REF) + "/"), "")
                for f in sorted(files):
                    if f.endswith(".py"):
                        full = os.path.join(d, f)
                        z.write(full, os.path.relpath(full, REF))
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
