"""The fixtures under tests/golden/ are outputs of the reference's own code (oracle/gen_golden.py).  Where the
reference tree is present (this container, not the GPU box) they are regenerated into a temporary directory and
compared array by array; everywhere, the full-set fixtures are checked against the sampled ones."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLD, ROOT


def _same(a, b):
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k


@pytest.mark.skipif(not Path("/root/reference/lib/layer.c").exists(), reason="reference tree absent")
def test_fixtures_regenerate_bit_for_bit(tmp_path):
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "oracle", "_ref/libqmann_ref.so", "joint"], check=True,
                   stdout=subprocess.DEVNULL)
    env = dict(os.environ, QMANN_GOLDEN_OUT=str(tmp_path))
    subprocess.run([sys.executable, str(ROOT / "oracle" / "gen_golden.py")], check=True, env=env, stdout=subprocess.DEVNULL)
    made = sorted(p.name for p in tmp_path.glob("*.npz"))
    kept = sorted(p.name for p in GOLD.glob("*.npz"))
    # babi_qa1_en1k_sets.npz holds dataset bytes for the reference-host GPU test; it is not a gen_golden output
    assert made == [k for k in kept if k != "babi_qa1_en1k_sets.npz"], (made, kept)
    for name in made:
        _same(np.load(tmp_path / name), np.load(GOLD / name))


def _bow(words, V, with_time):
    out = np.zeros((words.shape[0], V), np.float32)
    for r, row in enumerate(words):
        ent = [int(w) for w in row if w != 0xFF]
        if with_time and ent:
            out[r, ent.pop()] = 1.0
        for w in ent:
            out[r, w] += 1.0
    return out


def test_full_qa1_set_contains_the_sampled_fixture():
    full, part = np.load(GOLD / "babi_qa1_test1000_words.npz"), np.load(GOLD / "babi_qa1_test64.npz")
    assert len(full["n_sen"]) == 1000 and int(full["dim_input"]) == int(part["dim_input"])
    n = len(part["n_sen"])
    assert np.array_equal(full["n_sen"][:n], part["n_sen"])
    rows = int(part["n_sen"].sum())
    V = int(part["dim_input"])
    assert np.array_equal(_bow(full["story_words"][:rows], V, True), part["story"].astype(np.float32))
    assert np.array_equal(_bow(full["question_words"][:n], V, False), part["question"].astype(np.float32))
    assert np.array_equal(full["answer"][:n], part["answer"].argmax(1))


def test_full_joint_set_contains_the_sampled_fixture():
    full, part = np.load(GOLD / "babi_joint20_test20000_words.npz"), np.load(GOLD / "babi_joint20_test2000.npz")
    assert len(full["n_sen"]) == 20000 and int(full["dim_input"]) == int(part["dim_input"])
    per_f, per_p = 1000, 100
    offs_f = np.concatenate([[0], np.cumsum(full["n_sen"].astype(np.int64))])
    offs_p = np.concatenate([[0], np.cumsum(part["n_sen"].astype(np.int64))])
    w8 = lambda a, w: np.where(a[:, :w] == 0xFFFF, 0xFF, a[:, :w]).astype(np.uint8)
    ws, wq = full["story_words"].shape[1], full["question_words"].shape[1]
    for t in range(20):
        f0, p0 = t * per_f, t * per_p
        assert np.array_equal(full["n_sen"][f0:f0 + per_p], part["n_sen"][p0:p0 + per_p])
        assert np.array_equal(full["story_words"][offs_f[f0]:offs_f[f0 + per_p]], w8(part["story_words"][offs_p[p0]:offs_p[p0 + per_p]], ws))
        assert np.array_equal(full["question_words"][f0:f0 + per_p], w8(part["question_words"][p0:p0 + per_p], wq))
    assert (part["story_words"][:, ws:] == 0xFFFF).all() and (part["question_words"][:, wq:] == 0xFFFF).all()
