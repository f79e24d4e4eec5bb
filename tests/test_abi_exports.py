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
                + abi.header_symbols("qmann_weights.h") + abi.header_symbols("qmann_model.h"))
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
              + abi.header_symbols("qmann_weights.h") + abi.header_symbols("qmann_model.h")):
        assert s in names, s


def test_reference_objects_link_against_the_library():
    """oracle/_ref/libqmann_ref.so is the reference's unmodified lib/layer.c + lib/common.c
    (+ MemN2N/sample.c) linked with -z defs against libqmann_hip.so: if it exists and loads,
    all 56 cuda_* imports of layer.o were satisfied by this library."""
    so = ROOT / "oracle" / "_ref" / "libqmann_ref.so"
    if not so.exists():
        pytest.skip("reference build not present (built only where /root/reference exists)")
    L = ctypes.CDLL(str(so))
    assert hasattr(L, "dense_mat_fwd") and hasattr(L, "dot_mat_vec_fwd") and hasattr(L, "softmax_fwd")
    und = subprocess.run(["nm", "-D", "--undefined-only", str(so)], capture_output=True, text=True,
                         check=True).stdout
    cuda_imports = sorted({l.split()[-1] for l in und.splitlines() if " U cuda_" in l})
    assert len(cuda_imports) == 56, len(cuda_imports)


def test_lds_sizing_helper(abi):
    small = abi.lib.qmann_hops_lds_bytes(50)
    big = abi.lib.qmann_hops_lds_bytes(10000)
    assert small % 16 == 0 and big % 16 == 0
    assert big - small == 10000 - 64 + 0 or big > small
    assert big < 160 * 1024
