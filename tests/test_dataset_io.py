"""include/qmann_dataset.h: the reference's parsed bAbI record files -> uint16 word lists, natively.  Pinned against the
fixtures the reference's OWN sample.c produced (tests/golden/*_words.npz, made by oracle/gen_golden.py through
oracle/_ref): same dictionary, same dimensions, and for every sentence / question the same bag of words, the same time
entry, the same answer index.  Host-only: no GPU needed."""
import re
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLD, load_pkg

DATA = Path("/root/reference/MemN2N/dataset")
needs_reference = pytest.mark.skipif(not DATA.exists(), reason="the reference's dataset files live in /root/reference")


@pytest.fixture(scope="module")
def abi():
    load_pkg()
    import qmann_amd.abi as abi
    return abi


def compare_with_fixture(ds, g):
    assert ds["dim_input"] == int(g["dim_input"]) and ds["dim_dict"] == int(g["dim_dict"]) and ds["max_line"] == int(g["max_line"])
    n_sen = g["n_sen"].astype(np.int64)
    assert ds["n_query"] == len(n_sen)
    np.testing.assert_array_equal(np.diff(ds["row_off"].astype(np.int64)), n_sen)
    fs = np.where(g["story_words"] == 0xFF, 0xFFFF, g["story_words"].astype(np.uint16))
    fq = np.where(g["question_words"] == 0xFF, 0xFFFF, g["question_words"].astype(np.uint16))
    sw, qw = ds["story_words"], ds["question_words"]
    assert sw.shape[0] == fs.shape[0]
    for r in range(sw.shape[0]):
        mine = [int(w) for w in sw[r] if w != 0xFFFF]
        ref = [int(w) for w in fs[r] if w != 0xFFFF]
        assert mine[-1] == ref[-1], r                              # the time entry closes the list
        assert sorted(mine[:-1]) == ref[:-1], r                    # (the fixture lists a row's words by index)
    for q in range(qw.shape[0]):
        assert sorted(int(w) for w in qw[q] if w != 0xFFFF) == [int(w) for w in fq[q] if w != 0xFFFF], q
    fa = g["answer"].astype(np.int64)
    mine = ds["answer"].astype(np.int64)
    np.testing.assert_array_equal(np.where(mine == 0xFFFFFFFF, 0xFF, mine), fa)


@needs_reference
def test_qa1_test_set_equals_the_reference_vectorisation(abi):
    sub = DATA / "en_10k_parsed"
    ds = abi.load_dataset(sub / "qa1_single-supporting-fact_train_set", sub / "qa1_single-supporting-fact_test_set", 50)
    compare_with_fixture(ds, np.load(GOLD / "babi_qa1_test1000_words.npz"))
    # sentence order is kept (the fixture only knows bags of words): "John travelled to the hallway" -> 5 words + time
    assert (ds["story_words"][0] != 0xFFFF).sum() == 6 and ds["dim_word"] == 7


@needs_reference
def test_joint_20_tasks_equal_the_reference_vectorisation(abi, tmp_path):
    sub = DATA / "en_1k_parsed"
    tasks = sorted({re.sub(r"_(train|test)_set$", "", f.name) for f in sub.iterdir()}, key=lambda t: int(re.match(r"qa(\d+)_", t).group(1)))
    assert len(tasks) == 20
    for kind in ("train", "test"):                                  # the 20 files back to back, as oracle/gen_golden.py builds them
        bodies, total = [], 0
        for t in tasks:
            txt = (sub / f"{t}_{kind}_set").read_text()
            m = re.match(r"\n\+NS\+\n(\d+)\n\n", txt)
            total += int(m.group(1))
            bodies.append(txt[m.end():].rstrip("\n") + "\n\n")
        (tmp_path / f"joint_{kind}_set").write_text(f"\n+NS+\n{total}\n\n" + "".join(bodies))
    ds = abi.load_dataset(tmp_path / "joint_train_set", tmp_path / "joint_test_set", 64)
    compare_with_fixture(ds, np.load(GOLD / "babi_joint20_test20000_words.npz"))


def write_set(path, records):
    out = ["", "+NS+", str(len(records)), ""]
    for i, (sens, q, a) in enumerate(records):
        out += ["+I+", str(i), "+S+", str(len(sens))] + [s + " " for s in sens] + ["+Q+", q + " ", "+A+", a, ""]
    path.write_text("\n".join(out) + "\n")


def test_rules_on_a_hand_made_file(abi, tmp_path):
    """dictionary in order of first appearance without case, NULL at 0; the LAST max_len sentences of a long story; time
    indices count back from the most recent sentence; a test word outside the dictionary is dropped; sample caps"""
    train = [(["Mary went home", "John  went   out"], "Where is Mary", "home"),
             (["a b c d", "e"], "where IS john", "out")]
    test = [(["s1 x", "MARY went Home", "john went out", "e d c"], "Where is zebra", "OUT"),
            ([], "is", "nowhere"),
            (["a"], "a", "a")]
    write_set(tmp_path / "tr", train); write_set(tmp_path / "te", test)
    ds = abi.load_dataset(tmp_path / "tr", tmp_path / "te", 50)
    words = ["null", "mary", "went", "home", "john", "out", "where", "is", "a", "b", "c", "d", "e"]
    ix = {w: i for i, w in enumerate(words)}
    assert ds["dim_dict"] == len(words) and ds["max_line"] == 2 and ds["dim_word"] == 5 and ds["dim_input"] == len(words) + 2
    assert ds["n_query"] == 3 and list(ds["row_off"]) == [0, 2, 2, 3]            # story 0 is cut to its LAST two sentences
    row = lambda r: [int(w) for w in ds["story_words"][r] if w != 0xFFFF]
    assert row(0) == [ix["john"], ix["went"], ix["out"], len(words) + 1]          # older sentence: time index dim_dict + 1
    assert row(1) == [ix["e"], ix["d"], ix["c"], len(words) + 0]
    assert row(2) == [ix["a"], len(words) + 0]
    q = lambda i: [int(w) for w in ds["question_words"][i] if w != 0xFFFF]
    assert q(0) == [ix["where"], ix["is"]]                                        # "zebra" is not in the dictionary
    assert list(ds["answer"]) == [ix["out"], 0xFFFFFFFF, ix["a"]]
    assert ds["story_words"].shape[1] == 8                                        # pitch: a multiple of 4 slots
    capped = abi.load_dataset(tmp_path / "tr", tmp_path / "te", 50, n_train_cap=1, n_test_cap=2)
    assert capped["dim_dict"] == 8 and capped["n_query"] == 2


def test_missing_file_is_an_error_code(abi, tmp_path):
    import ctypes as C
    ds = abi.Dataset()
    assert abi.lib.qmann_dataset_load(str(tmp_path / "nope").encode(), b"/dev/null", 50, 0, 0, C.byref(ds)) == abi.QMANN_EIO
    (tmp_path / "bad").write_text("garbage\n")
    assert abi.lib.qmann_dataset_load(str(tmp_path / "bad").encode(), str(tmp_path / "bad").encode(), 50, 0, 0, C.byref(ds)) == abi.QMANN_EIO


def test_unknown_question_word_keeps_the_positions_of_the_others(abi, tmp_path):
    """EN_PE weighs a question word by its POSITION in the question (sample.c:559: pe_w[word][k]); an unknown word must leave a
    0xFFFF hole in its slot, not shift the words after it"""
    train = [(["mary went home now"], "where is mary now", "home")]          # dim_word 5: a question keeps 4 words
    test = [(["mary went home"], "where zebra mary now", "home")]
    write_set(tmp_path / "tr", train); write_set(tmp_path / "te", test)
    ds = abi.load_dataset(tmp_path / "tr", tmp_path / "te", 50)
    words = ["null", "mary", "went", "home", "now", "where", "is"]
    ix = {w: i for i, w in enumerate(words)}
    assert [int(w) for w in ds["question_words"][0][:4]] == [ix["where"], 0xFFFF, ix["mary"], ix["now"]]


@pytest.mark.parametrize("where", ["samples", "sentences"])
def test_count_line_that_is_no_number_is_a_format_error(abi, tmp_path, where):
    import ctypes as C
    good = "\n+NS+\n1\n\n+I+\n0\n+S+\n1\nmary went home \n+Q+\nwhere is mary \n+A+\nhome\n\n"
    (tmp_path / "good").write_text(good)
    bad = good.replace("+NS+\n1\n", "+NS+\nmany\n") if where == "samples" else good.replace("+S+\n1\n", "+S+\n\n")
    (tmp_path / "bad").write_text(bad)
    ds = abi.Dataset()
    g, b = str(tmp_path / "good").encode(), str(tmp_path / "bad").encode()
    assert abi.lib.qmann_dataset_load(g, g, 50, 0, 0, C.byref(ds)) == 0
    abi.lib.qmann_dataset_free(C.byref(ds))
    assert abi.lib.qmann_dataset_load(b, g, 50, 0, 0, C.byref(ds)) == abi.QMANN_EIO
    assert abi.lib.qmann_dataset_load(g, b, 50, 0, 0, C.byref(ds)) == abi.QMANN_EIO
