"""Build every native artefact in-tree.

  q-mann_amd/lib/libqmann_hip.so   HIP kernels + C-ABI (hipcc, gfx950)
  oracle/libqmann_oracle.so        CPU restatement (gcc)  -- test infrastructure
  oracle/_ref/libqmann_ref.so      the reference's own C sources, compiled where
                                   they lie under /root/reference (only when that
                                   tree is present)      -- test infrastructure

hipcc cross-compiles for gfx950 without a GPU.  Nothing here runs a kernel.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "q-mann_amd"
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
HIP_LIB = LIBDIR / "libqmann_hip.so"
ORACLE_DIR = ROOT / "oracle"
ORACLE_LIB = ORACLE_DIR / "libqmann_oracle.so"
REF_DIR = Path("/root/reference")
REF_LIB = ORACLE_DIR / "_ref" / "libqmann_ref.so"

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
    "-ffp-contract=off",            # float-on-grid arithmetic must not be fused
    "-Wall", "-Wno-unused-function",
]


def _run(cmd, **kw):
    print("+", " ".join(str(c) for c in cmd), flush=True)
    subprocess.run([str(c) for c in cmd], check=True, **kw)


def _stale(target: Path, sources) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(s).stat().st_mtime > t for s in sources)


def hip_sources():
    return sorted(CSRC.glob("*.hip"))


def build_hip(force: bool = False) -> Path:
    srcs = hip_sources()
    deps = srcs + sorted(CSRC.glob("*.h")) + sorted((ROOT / "include").glob("*.h"))
    LIBDIR.mkdir(exist_ok=True)
    objdir = LIBDIR / "obj"
    objdir.mkdir(exist_ok=True)
    objs = []
    hdrs = [d for d in deps if d.suffix == ".h"]
    for s in srcs:
        o = objdir / (s.stem + ".o")
        if force or _stale(o, [s] + hdrs):
            _run([HIPCC, *HIP_FLAGS, "-I", ROOT / "include", "-c", s, "-o", o])
        objs.append(o)
    if force or _stale(HIP_LIB, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB, *objs])
    return HIP_LIB


def build_oracle(force: bool = False) -> Path:
    _run(["make", "-C", ORACLE_DIR, "oracle"] + (["-B"] if force else []))
    return ORACLE_LIB


def build_ref(force: bool = False):
    """Reference C sources compiled in place; skipped when the tree is absent (GPU box)."""
    if not REF_DIR.exists():
        return REF_LIB if REF_LIB.exists() else None
    _run(["make", "-C", ORACLE_DIR, "ref", f"HIP_LIB={HIP_LIB}"] + (["-B"] if force else []))
    return REF_LIB


def build_all(force: bool = False):
    build_hip(force)
    build_oracle(force)
    build_ref(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
