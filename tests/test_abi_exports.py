"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and
exports, unmangled, every entry point the headers declare.  No compute calls here."""
import ctypes
import subprocess

import pytest

from conftest import ROOT, load_pkg


@pytest.fixture(scope="module")
def abi():
    load_pkg()
    import qmann_amd.abi as a
    return a


def test_library_loads_and_reports_66(abi):
    assert abi.lib.qmann_abi_symbol_count() == 66


def test_every_declared_symbol_is_exported(abi):
    declared = (abi.header_symbols("qmann_abi.h") + abi.header_symbols("qmann_batch.h")
                + abi.header_symbols("qmann_weights.h") + abi.header_symbols("qmann_model.h")
                + abi.header_symbols("qmann_dataset.h") + abi.header_symbols("qmann_dist.h"))
    assert "qmann_comm_broadcast_params" in declared and "qmann_model_create_from_params" in declared
    assert "qmann_weights_save" in declared and "qmann_weights_load" in declared
    assert "qmann_model_create" in declared and "qmann_model_forward_words" in declared
    cuda = [s for s in declared if s.startswith("cuda_")]
    assert len(cuda) == 66, len(cuda)
    missing = [s for s in declared if not hasattr(abi.lib, s)]
    assert not missing, missing


def test_symbols_are_unmangled_c(abi):
    out = subprocess.run(["nm", "-D", "--defined-only", str(abi.LIB_PATH)], capture_output=True, text=True,
                         check=True).stdout
    names = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for s in (abi.header_symbols("qmann_abi.h") + abi.header_symbols("qmann_batch.h")
              + abi.header_symbols("qmann_weights.h") + abi.header_symbols("qmann_model.h")
                + abi.header_symbols("qmann_dataset.h") + abi.header_symbols("qmann_dist.h")):
        assert s in names, s


def _nm_undefined(path):
    und = subprocess.run(["nm", "-D", "--undefined-only", str(path)], capture_output=True, text=True, check=True).stdout
    return sorted({l.split()[-1] for l in und.splitlines() if " U " in l})


def test_reference_objects_link_against_the_library(abi):
    """oracle/_ref/MemN2N_ref is the reference's WHOLE unmodified host program (MemN2N.c + sample.c + layer.c +
    common.c compiled where they lie) linked against libqmann_hip.so in place of layer_cuda.o / libcudart: the
    executable exists only if every cuda_* import of layer.o (56) and MemN2N.o (10) was satisfied by this library."""
    exe = ROOT / "oracle" / "_ref" / "MemN2N_ref"
    if not exe.exists():
        pytest.skip("reference build not present (built only where /root/reference exists)")
    cuda_imports = [s for s in _nm_undefined(exe) if s.startswith("cuda_")]
    declared = [s for s in abi.header_symbols("qmann_abi.h") if s.startswith("cuda_")]
    # (the stock define.h leaves one of MemN2N.o's ten helpers, cuda_copy_dev2host, unreferenced after preprocessing)
    assert len(cuda_imports) >= 65 and set(cuda_imports) <= set(declared), sorted(set(cuda_imports) - set(declared))
    needed = subprocess.run(["readelf", "-d", str(exe)], capture_output=True, text=True, check=True).stdout
    assert "libqmann_hip.so" in needed


def test_fixture_generator_holds_no_product_code():
    """The reference's CPU code that pins the oracle (libqmann_ref*.so) and serves as the CPU baseline
    (libqmann_refcpu_*.so) is reference objects + glue and nothing else: it depends on no product library, DEFINES no
    cuda_* symbol (no stand-ins for the CUDA object: layer.o's imports stay undefined and, with en_gpu_model = false,
    uncalled) and loads with lazy binding."""
    import sys
    sys.path.insert(0, str(ROOT / "oracle"))
    from pyoracle import load_lazy
    ref = ROOT / "oracle" / "_ref"
    libs = sorted(ref.glob("libqmann_ref*.so"))
    if not libs:
        pytest.skip("reference build not present (built only where /root/reference exists)")
    assert not (ROOT / "oracle" / "cuda_stubs.c").exists()
    for so in libs:
        dyn = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True, check=True).stdout
        assert not [l for l in dyn.splitlines() if l.split()[-1].startswith("cuda_")], so
        needed = subprocess.run(["readelf", "-d", str(so)], capture_output=True, text=True, check=True).stdout
        assert "qmann_hip" not in needed and "amdhip" not in needed, so
        # what `-z defs` used to guarantee at link time, now that the cuda_* imports stay undefined on purpose (`-z lazy`):
        # every OTHER undefined symbol resolves in libc / libm / libpthread (nm prints those with their @GLIBC version) -- a
        # misspelt or missing symbol of ref_glue.c / ref_forward.c would show up here unversioned instead of ending the process
        # at its first call
        stray = [u for u in _nm_undefined(so) if not u.startswith("cuda_") and "@GLIBC" not in u]
        assert not stray, (so, stray)
        L = load_lazy(so)
        assert hasattr(L, "dense_mat_fwd") and hasattr(L, "softmax_fwd") and hasattr(L, "hamming_similarity")


def test_lds_sizing_helper(abi):
    small = abi.lib.qmann_hops_lds_bytes(50)
    big = abi.lib.qmann_hops_lds_bytes(10000)
    assert small % 16 == 0 and big % 16 == 0
    assert big - small == 10000 - 64 + 0 or big > small
    assert big < 160 * 1024


def test_rccl_is_not_a_load_time_dependency(abi):
    """librccl (0.5 GB) is loaded with dlopen by the first qmann_comm_* call, never by loading the library: the reference's
    single-GPU host links the drop-in verbs and must not pay for it"""
    needed = subprocess.run(["readelf", "-d", str(abi.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    assert "rccl" not in needed and "nccl" not in needed
    und = _nm_undefined(abi.LIB_PATH)
    assert not [s for s in und if s.startswith("nccl")], und


def test_shard_ranges_tile_the_batch(abi):
    """qmann_shard_range (C) = the contiguous split the Python plumbing uses; ranges tile [0, n) in rank order"""
    from qmann_amd.parallel import shard_range
    for n in (0, 1, 7, 1000, 65536, 65537):
        for world in (1, 2, 3, 8):
            got = [abi.shard_range(n, r, world) for r in range(world)]
            assert got == [shard_range(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n and all(got[i][1] == got[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in got]
            assert max(sizes) - min(sizes) <= 1
