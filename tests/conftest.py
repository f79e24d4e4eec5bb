"""Shared fixtures.  `-m "not gpu"` runs here (no GPU); `-m gpu` runs on an MI355X box."""
from __future__ import annotations

import importlib.util
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Import the product package (directory `q-mann_amd/`, module name `qmann_amd`)."""
    if "qmann_amd" in sys.modules:
        return sys.modules["qmann_amd"]
    pkg_dir = ROOT / "q-mann_amd"
    spec = importlib.util.spec_from_file_location("qmann_amd", pkg_dir / "__init__.py",
                                                  submodule_search_locations=[str(pkg_dir)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["qmann_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(autouse=True)
def _tuning_switches_follow_the_environment():
    """The library reads its QMANN_* A/B switches once per process (csrc/tuning.hip).  A test that sets one calls
    qmann_tuning_reload() itself; this puts the library back in step with the restored environment afterwards (set up
    before `monkeypatch`, so it is finalised after monkeypatch has undone its changes)."""
    yield
    abi = sys.modules.get("qmann_amd.abi")
    if abi is not None:
        abi.lib.qmann_tuning_reload()


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def oracle():
    from pyoracle import Oracle, ORACLE_SO
    if not ORACLE_SO.exists():
        import subprocess
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "oracle"], check=True)
    return Oracle()


@pytest.fixture(scope="session")
def gold():
    def _load(name):
        return np.load(GOLD / name)
    return _load
