"""tests/golden/trained_qa1 (CPU side): the weight files the reference's host program trained through this library load with the
dimensions the record files give, are tied across the hops as the program leaves them, and the ORACLE alone
reproduces the test error the reference program printed (the GPU side is tests/test_gpu_words.py)."""
import json

import numpy as np

from conftest import GOLD, load_pkg


def _bow(words, V, with_time):
    out = np.zeros((words.shape[0], V), np.float32)
    for r, row in enumerate(words):
        ent = [int(w) for w in row if w != 0xFFFF]
        t = ent.pop() if (with_time and ent) else None
        for w in ent:
            out[r, w] += 1.0
        if t is not None:
            out[r, t] = 1.0
    return out


def test_oracle_reproduces_the_reference_programs_test_error(oracle, tmp_path):
    load_pkg()
    import qmann_amd.abi as abi
    tdir = GOLD / "trained_qa1"
    rec = json.loads((tdir / "reference_run.json").read_text())
    assert rec["verify_line"].endswith("equal") and rec["epochs"] == 100
    assert rec["train_error_last_epoch"] < rec["train_error_first_epoch"] - 0.3        # it did learn
    g = np.load(GOLD / "babi_qa1_en1k_sets.npz")
    (tmp_path / "train").write_bytes(g["train_set"].tobytes()); (tmp_path / "test").write_bytes(g["test_set"].tobytes())
    ds = abi.load_dataset(tmp_path / "train", tmp_path / "test", 50)
    V, D, H, iwl = ds["dim_input"], 60, 3, int(rec["argv"][3])
    frac = 7 - iwl
    fmt = [(iwl, frac)] * H
    fmt_w = [(iwl + 1, frac - 1), (iwl, frac), (iwl - 1, frac + 1)]                    # EN_MQ (MemN2N.c:748-754)
    cfg = dict(n_hop=H, dim_emb=D, dim_input=V, attention_mode=2, softmax_variant=0, f_fixed=True, en_lin_map=True, fmt=fmt,
               fmt_w=fmt_w, fmt_att=list(fmt), fmt_bin=(iwl, frac))

    def rd(name, shape):                                                               # column-major files (qmann_weights.h)
        a = np.fromfile(tdir / name, np.float32)
        assert a.size == int(np.prod(shape)), (name, a.size, shape)
        return a
    wa = rd("w_emb_a_float.bin", (H, V, D)).reshape(H, V, D).transpose(0, 2, 1)
    wc = rd("w_emb_c_float.bin", (H, V, D)).reshape(H, V, D).transpose(0, 2, 1)
    wq = rd("w_emb_q_float.bin", (V, D)).reshape(V, D).T
    wh = rd("w_lin_map_float.bin", (H, D, D)).reshape(H, D, D).transpose(0, 2, 1)
    wans = rd("w_float.bin", (D, V)).reshape(D, V).T
    # layer-wise weight tying: after every update hop 0's embedding matrices are copied over the other hops' (MemN2N.c:1770-1773);
    # the matrices themselves are float masters, every layer quantises them on the way in
    assert np.array_equal(wa[1], wa[0]) and np.array_equal(wa[2], wa[0]) and np.array_equal(wc[1], wc[0])
    wts = dict(w_q=np.ascontiguousarray(wq), w_a=[np.ascontiguousarray(x) for x in wa], w_c=[np.ascontiguousarray(x) for x in wc],
               w_h=[np.ascontiguousarray(x) for x in wh], w_ans=np.ascontiguousarray(wans))
    m = oracle.make_model(cfg, wts)
    pred, _, gap, _ = oracle.forward_words_batch(m, ds["story_words"], ds["question_words"], ds["row_off"])
    match = int((pred.astype(np.int64) == ds["answer"].astype(np.int64)).sum())
    unclear = int((gap <= 1e-6).sum())
    # the program's count, up to output near-ties the float tolerance may resolve either way (none expected)
    assert abs((1.0 - match / ds["n_query"]) - rec["err_test_result_csv"]) <= unclear / ds["n_query"] + 1e-9, (match, unclear)
