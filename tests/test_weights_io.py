"""Weight files (include/qmann_weights.h): host-only, runs without a GPU.
Layout facts checked here are the ones of the reference's disabled blocks MemN2N/MemN2N.c:2553-2618 /
:2853-2978 -- column-major matrices, hops back to back, sign-magnitude int32 words in the *_fixed files."""
import sys

import numpy as np
import pytest

from conftest import ROOT, load_pkg


@pytest.fixture(scope="module")
def model():
    load_pkg()
    import qmann_amd.model as m
    return m


@pytest.fixture(scope="module")
def oracle():
    sys.path.insert(0, str(ROOT / "oracle"))
    from pyoracle import Oracle
    return Oracle()


def make(H, D, V, seed):
    rng = np.random.default_rng(seed)
    return {"w_q": rng.normal(0, 1.5, (D, V)).astype(np.float32),
            "w_a": [rng.normal(0, 1.5, (D, V)).astype(np.float32) for _ in range(H)],
            "w_c": [rng.normal(0, 1.5, (D, V)).astype(np.float32) for _ in range(H)],
            "w_h": [rng.normal(0, 1.0, (D, D)).astype(np.float32) for _ in range(H)],
            "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32)}


def test_float_files_layout_and_round_trip(model, tmp_path):
    H, D, V = 3, 60, 30
    cfg = model.babi_cfg(V, 2, 0)
    w = make(H, D, V, 1)
    w["w_a"][1][2, 5] = -0.0                                    # bit patterns survive, not just values
    model.save_weights(tmp_path, w, cfg, fixed=False)
    raw = np.fromfile(tmp_path / "w_emb_a_float.bin", np.float32)
    assert raw.size == H * D * V
    # column-major per hop: element (i = row of w_mat, j = column) of hop h sits at h.D.V + j.D + i
    np.testing.assert_array_equal(raw.reshape(H, V, D), np.stack([m.T for m in w["w_a"]]))
    np.testing.assert_array_equal(np.fromfile(tmp_path / "w_float.bin", np.float32).reshape(D, V), w["w_ans"].T)
    assert np.fromfile(tmp_path / "w_emb_q_float.bin", np.float32).size == D * V
    assert np.fromfile(tmp_path / "w_lin_map_float.bin", np.float32).size == H * D * D
    assert not (tmp_path / "w_emb_a_fixed.bin").exists()
    back = model.load_weights(tmp_path, cfg)
    for k in ("w_q", "w_ans"):
        assert back[k].tobytes() == w[k].tobytes()
    for k in ("w_a", "w_c", "w_h"):
        for h in range(H):
            assert back[k][h].tobytes() == w[k][h].tobytes()


def test_fixed_files_are_reference_words_and_decode_to_quantised_weights(model, oracle, tmp_path):
    H, D, V = 3, 60, 30
    cfg = model.babi_cfg(V, 2, 0, iwl=2)                        # EN_MQ formats Q3.4 / Q2.5 / Q1.6
    w = make(H, D, V, 2)
    w["w_c"][0][0, 0] = -0.001                                  # truncates to "minus zero": 0x80000000
    w["w_c"][0][0, 1] = 100.0                                   # saturates
    model.save_weights(tmp_path, w, cfg, fixed=True)
    words = np.fromfile(tmp_path / "w_emb_c_fixed.bin", np.uint32).reshape(H, V, D)
    for h in range(H):
        iwl, frac = cfg["fmt_w"][h]
        want = np.array([[oracle.float2fixed(float(x), iwl, frac) for x in row] for row in w["w_c"][h].T],
                        np.int64).astype(np.uint32)
        np.testing.assert_array_equal(words[h], want)
    assert words[0, 0, 0] == 0x80000000
    assert words[0, 1, 0] == (1 << 7) - 1
    qw = np.fromfile(tmp_path / "w_emb_q_fixed.bin", np.uint32).reshape(V, D)
    iwl0, frac0 = cfg["fmt_w"][0]
    assert int(qw[3, 7]) == int(oracle.float2fixed(float(w["w_q"][7, 3]), iwl0, frac0)) & 0xFFFFFFFF
    back = model.load_weights(tmp_path, cfg, from_fixed=True)
    for h in range(H):
        iwl, frac = cfg["fmt_w"][h]
        for k in ("w_a", "w_c", "w_h"):
            want = np.vectorize(lambda x: oracle.quant(float(x), iwl, frac))(w[k][h]).astype(np.float32)
            np.testing.assert_array_equal(back[k][h], want)
    np.testing.assert_array_equal(back["w_ans"], w["w_ans"])    # the answer layer is float: read from w_float.bin
    assert np.signbit(back["w_c"][0][0, 0]) and back["w_c"][0][0, 0] == 0.0


def test_wrong_size_or_missing_file_is_an_error(model, tmp_path):
    H, D, V = 3, 60, 30
    cfg = model.babi_cfg(V, 2, 0)
    with pytest.raises(RuntimeError):
        model.load_weights(tmp_path, cfg)                       # nothing there
    model.save_weights(tmp_path, make(H, D, V, 3), cfg)
    with open(tmp_path / "w_emb_c_float.bin", "ab") as f:
        f.write(b"\\0\\0\\0\\0")
    with pytest.raises(RuntimeError):
        model.load_weights(tmp_path, cfg)
    cfg2 = model.babi_cfg(V + 1, 2, 0)                          # other dictionary size: sizes disagree
    with pytest.raises(RuntimeError):
        model.load_weights(tmp_path, cfg2)
