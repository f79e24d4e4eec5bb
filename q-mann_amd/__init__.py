"""qmann_amd -- MI355X-native quantized MemN2N inference path (Q-MANN test-phase forward).

The product is the C-ABI shared library `lib/libqmann_hip.so` (hand-written HIP
for gfx950; headers in /include).  This Python package is only the host-side
plumbing used by tests and bench.py: it loads the library with ctypes and hands
it raw device pointers (torch is used for device memory and torch.distributed,
nothing else).  There is no CPU fallback: if the library is missing or cannot be
loaded, importing `qmann_amd.abi` raises.

The directory is named `q-mann_amd`; import it through `load_pkg()` in
tests/conftest.py / bench.py (module name `qmann_amd`).
"""
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent
# QMANN_LIB_PATH: another build of the same library (A/B timing of two builds inside one GPU session)
LIB_PATH = Path(os.environ["QMANN_LIB_PATH"]) if os.environ.get("QMANN_LIB_PATH") else PKG_DIR / "lib" / "libqmann_hip.so"



def kernel_sources_sha16() -> str:
    """First 16 hex digits of the SHA-256 over the kernel sources (csrc/*.hip, csrc/*.h, in name order).  Counter figures
    taken in a separate profiler pass (profiles/traffic.json, profiles/mfma.json) are stamped with it; bench.py quotes them
    only while the sources they were measured on are the ones that built the loaded library."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted((PKG_DIR / "csrc").glob("*.h*")):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


__all__ = ["PKG_DIR", "ROOT", "LIB_PATH", "kernel_sources_sha16"]
