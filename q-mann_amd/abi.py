"""ctypes binding of libqmann_hip.so (the C-ABI product).  No fallback: a missing or
unloadable library is an ImportError here, and every op below goes through it."""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

from . import LIB_PATH, ROOT

if not LIB_PATH.exists():
    raise ImportError(
        f"{LIB_PATH} is missing -- build it with `python q-mann_amd/build.py` "
        "(there is no CPU or PyTorch fallback for the HIP path)")

lib = C.CDLL(str(LIB_PATH))

QMANN_MAX_HOP = 8
ATT_FLOAT, ATT_FIXED, ATT_APPX, ATT_HAMMING_V0, ATT_HAMMING_V1, ATT_SIGN = 1, 2, 3, 10, 11, 12
SOFTMAX_EXP, SOFTMAX_POW2 = 0, 1
CODE_TWOS, CODE_SIGNMAG = 0, 1

_vp = C.c_void_p
_u = C.c_uint
_b = C.c_bool


class Fmt(C.Structure):
    _fields_ = [("iwl", C.c_uint32), ("frac", C.c_uint32)]


class Net(C.Structure):
    _fields_ = [
        ("n_hop", C.c_uint32), ("dim_emb", C.c_uint32), ("dim_emb_pad", C.c_uint32), ("dim_input", C.c_uint32),
        ("attention_mode", C.c_uint32), ("softmax_base", C.c_uint32), ("en_lin_map", C.c_uint32),
        ("num_bit", C.c_uint32),
        ("act", Fmt * QMANN_MAX_HOP), ("w", Fmt * QMANN_MAX_HOP), ("att", Fmt * QMANN_MAX_HOP), ("bin", Fmt),
        ("lin_map", _vp * QMANN_MAX_HOP),
        ("softmax_shift_based", C.c_uint32), ("en_att_scale", C.c_uint32), ("att_scale", C.c_float * QMANN_MAX_HOP),
        ("en_non_linearity", C.c_uint32),
        ("en_pe", C.c_uint32), ("pe_dim_word", C.c_uint32),
    ]


class Weights(C.Structure):
    """include/qmann_weights.h: host matrices, row-major."""
    _fields_ = [
        ("n_hop", C.c_uint32), ("dim_emb", C.c_uint32), ("dim_input", C.c_uint32),
        ("w_q", _vp), ("w_a", _vp * QMANN_MAX_HOP), ("w_c", _vp * QMANN_MAX_HOP), ("w_h", _vp * QMANN_MAX_HOP),
        ("w_ans", _vp),
    ]


class Taps(C.Structure):
    _fields_ = [("score_codes", _vp), ("scores", _vp), ("probs", _vp), ("o", _vp), ("u", _vp)]


def header_symbols(header: str):
    """Function names declared in include/<header> (used by the export test)."""
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:cuda|qmann)_[a-z0-9_]+)\s*\(", text)))


def _proto(name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


# ---- boundary B (include/qmann_abi.h): the forward / utility verbs the tests drive ----
_pp = C.POINTER(_vp)
_proto("cuda_dot_mat_vec_constructor", None, [_pp, _pp, _pp, _pp, _pp, _u, _u, _b])
_proto("cuda_dot_mat_vec_init", None, [_vp] * 5 + [_u, _u, _b])
_proto("cuda_dot_mat_vec_fwd", None, [_vp] * 4 + [_u, _u, _b, _b, _u, _u, _u, _u, _u, _b])
_proto("cuda_dot_mat_vec_fwd_appx", None, [_vp] * 5 + [_u, _u, _b, _u, _u, _u, _u, _b, _b])
_proto("cuda_dot_mat_vec_destructor", None, [_vp] * 5)
_proto("cuda_softmax_constructor", None, [_pp, _pp, _pp, _u])
_proto("cuda_softmax_init", None, [_vp, _vp, _vp, _u])
_proto("cuda_softmax_fwd", None, [_vp] * 5 + [_u, _b, _b])
_proto("cuda_softmax_destructor", None, [_vp] * 3)
_proto("cuda_sum_vec_constructor", None, [_pp, _pp, _u])
_proto("cuda_sum_vec_fwd", None, [_vp] * 3 + [_u, _b, _u, _u, _u, _b])
_proto("cuda_sum_vec_destructor", None, [_vp] * 2)
_proto("cuda_dense_constructor", None, [_pp] * 10 + [_u, _u])
_proto("cuda_dense_init", None, [_vp] * 9 + [_u, _u])
_proto("cuda_dense_fwd", None, [_vp] * 5 + [_u, _u, C.c_char_p, _b, _u, _u, _u, _u, _u, _b])
_proto("cuda_dense_destructor", None, [_vp] * 8)
_proto("cuda_dense_mat_constructor", None, [_pp] * 10 + [_u, _u, _u])
_proto("cuda_dense_mat_init", None, [_vp] * 9 + [_u, _u, _u])
_proto("cuda_dense_mat_fwd", None, [_vp] * 5 + [_u, _u, _u, _b, _u, _u, _u, _b])
_proto("cuda_dense_mat_destructor", None, [_vp] * 7)
_proto("cuda_cross_entropy_constructor", None, [_pp] * 8 + [_u])
_proto("cuda_cross_entropy_init", None, [_vp] * 7 + [_u])
_proto("cuda_cross_entropy_run", None, [_vp] * 14 + [_u, _u])
_proto("cuda_cross_entropy_cost_load", None, [_vp] * 6)
_proto("cuda_cross_entropy_m_cnt_load", None, [_vp] * 6)
_proto("cuda_cross_entropy_destructor", None, [_vp] * 8)
_proto("cuda_activation_fwd", None, [_vp, _vp, C.c_char_p, _u, _b, _u, _u, _u])
_proto("cuda_scale_fwd", None, [_vp, _vp, _vp, _u, _b, _u, _u, _u, _b])
_proto("cuda_data_constructor", None, [_pp, _pp, _pp, _u, _u, _u])
_proto("cuda_data_in", None, [_vp] * 6 + [_u, _u, _u])
_proto("cuda_data_destructor", None, [_vp] * 3)
_proto("cuda_copy_mat", None, [_vp, _vp, _u, _u, _b])
_proto("cuda_accum_mat", None, [_vp, _vp, _u, _u, _b])
_proto("cuda_set_value", None, [_vp, C.c_float, _u, _u, _u])
_proto("cuda_copy_dev2host", None, [_vp, _vp, _u])
_f = C.c_float
_proto("cuda_dot_mat_vec_bwd", None, [_vp] * 6 + [_u, _u, _b, _b, _u, _u, _u, _u, _u, _b])
_proto("cuda_dot_mat_vec_bwd_appx", None, [_vp] * 7 + [_u, _u, _b, _u, _u, _u, _u, _b, _b, _u])
_proto("cuda_softmax_bwd", None, [_vp] * 4 + [_u, _b, _b])
_proto("cuda_sum_vec_bwd", None, [_vp] * 4 + [_u])
_proto("cuda_dense_bwd", None, [_vp] * 9 + [_u, _u, C.c_char_p, _b, _u, _u, _u, _u, _u, _b])
_proto("cuda_dense_w_up", None, [_vp] * 6 + [_u, _u, _u, C.POINTER(_f), C.POINTER(_f), C.POINTER(_f), _b, _u, _u, _u, _b])
_proto("cuda_dense_mat_bwd", None, [_vp] * 8 + [_u, _u, _u, _b, _u, _u, _u, _b])
_proto("cuda_dense_mat_w_up", None, [_vp] * 8 + [_u, _u, _u, C.POINTER(_f), C.POINTER(_f), C.POINTER(_f), _b, _u, _u, _u, _b])
_proto("cuda_dup_grad_bwd", None, [_vp] * 4 + [_u, _b, _u, _u, _u])
_proto("cuda_activation_bwd", None, [_vp, _vp, _vp, C.c_char_p, _u, _b, _u, _u, _u])
_proto("qmann_abi_set_softmax_base", None, [C.c_int])
_proto("qmann_abi_symbol_count", _u, [])


class DeferStats(C.Structure):
    """include/qmann_abi.h: qmann_defer_stats"""
    _fields_ = [(n, C.c_ulonglong) for n in ("ops_queued", "ops_replayed", "queries_batched", "batches", "models_built",
                                             "verify_runs", "verify_mismatch")] + [(n, C.c_double) for n in ("ms_batched", "ms_replayed", "ms_model")]


_proto("qmann_abi_set_defer", None, [C.c_int])
_proto("qmann_abi_flush", None, [])
_proto("qmann_abi_invalidate_model", None, [])
_proto("qmann_abi_defer_stats", None, [C.POINTER(DeferStats)])
# The tests and bench.py drive single verbs and then look at the device buffers through torch, not through a cuda_* verb:
# for this plumbing the forward verbs launch at once (the deferred queue is the default for C hosts; tests of the queue
# itself switch it on again or run the reference's host programs as subprocesses).
lib.qmann_abi_set_defer(0)


def defer_stats() -> dict:
    st = DeferStats()
    lib.qmann_abi_defer_stats(C.byref(st))
    return {n: getattr(st, n) for n, _ in DeferStats._fields_}

# ---- batched int8 API (include/qmann_batch.h) ----
_proto("qmann_hops_lds_bytes", C.c_size_t, [C.c_uint32])
_proto("qmann_tuning_reload", None, [])
_proto("qmann_check_slots", C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp, _vp])
_proto("qmann_quantize_i8", C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_uint32, Fmt, C.c_int, _vp])
_proto("qmann_hops_i8", C.c_int, [C.POINTER(Net), _vp, _vp, C.c_size_t, _vp, C.c_uint32, _vp, _vp,
                                  C.POINTER(Taps), C.c_uint32, _vp])
_proto("qmann_pack_bitplanes", C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_uint32, _vp])
_proto("qmann_hops_packed", C.c_int, [C.POINTER(Net), _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_uint32, _vp, _vp,
                                      C.POINTER(Taps), C.c_uint32, _vp])
_proto("qmann_answer_f32", C.c_int, [C.POINTER(Net), _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp])
_proto("qmann_answer_f32_serial", C.c_int, [C.POINTER(Net), _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp])
_proto("qmann_answer_i8", C.c_int, [C.POINTER(Net), _vp, Fmt, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp])
_proto("qmann_embed_story", C.c_int, [C.POINTER(Net), _vp, C.c_uint32, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp,
                                      C.c_size_t, _vp])
_proto("qmann_quantize_table_i8", C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, Fmt, _vp])
_proto("qmann_embed_story_idx", C.c_int, [C.POINTER(Net), _vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_vp),
                                          C.POINTER(_vp), _vp, _vp, C.c_size_t, _vp])
_proto("qmann_embed_query_idx", C.c_int, [C.POINTER(Net), _vp, C.c_uint32, _vp, _vp, C.c_uint32, _vp])
_proto("qmann_embed_query", C.c_int, [C.POINTER(Net), _vp, _vp, _vp, C.c_uint32, _vp])
_proto("qmann_bow_to_words", C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp])
_proto("qmann_embed_story_rows", C.c_int, [C.POINTER(Net), _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp])
_proto("qmann_embed_query_rows", C.c_int, [C.POINTER(Net), _vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp])
_proto("qmann_model_create", C.c_int, [C.POINTER(_vp), C.POINTER(Net), C.POINTER(Weights), _vp])
_proto("qmann_model_destroy", None, [_vp])
_proto("qmann_model_create_on", C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(Net), C.POINTER(Weights), _vp])
_proto("qmann_model_device", C.c_int, [_vp])
_proto("qmann_model_params", C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)])
_proto("qmann_model_create_from_params", C.c_int, [C.POINTER(_vp), C.c_int, _vp, C.c_size_t, _vp])
_proto("qmann_params_validate", C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(Net)])
_proto("qmann_model_net", C.c_int, [_vp, C.POINTER(Net), C.POINTER(_vp)])
_proto("qmann_dequantize_table_f32", C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, Fmt, _vp])
# ---- include/qmann_dist.h: shards, RCCL rendezvous, parameter broadcast ----
COMM_ID_BYTES = 128
_proto("qmann_shard_range", None, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)])
_proto("qmann_comm_probe", C.c_int, [C.c_int])
_proto("qmann_comm_get_id", C.c_int, [_vp])
_proto("qmann_comm_init_rank", C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, _vp, C.c_int])
_proto("qmann_comm_destroy", None, [_vp])
_proto("qmann_comm_info", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)])
_proto("qmann_comm_broadcast_params", C.c_int, [_vp, C.c_int, _vp, C.POINTER(_vp), C.POINTER(C.c_size_t), _vp])
_proto("qmann_params_free", None, [_vp])
_proto("qmann_comm_broadcast", C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp])
_proto("qmann_comm_allgather_u32", C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp])
_proto("qmann_model_forward_words", C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32, _vp, C.c_uint32,
                                              C.c_uint32, _vp, _vp, _vp, _vp, _vp])
_proto("qmann_model_forward_bow", C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, _vp])
_proto("qmann_model_last_u", _vp, [_vp])
_proto("qmann_weights_save", C.c_int, [C.c_char_p, C.POINTER(Weights), C.POINTER(Fmt)])
_proto("qmann_weights_load", C.c_int, [C.c_char_p, C.POINTER(Weights), C.c_int, C.POINTER(Fmt)])


class Dataset(C.Structure):
    """include/qmann_dataset.h"""
    _fields_ = [("n_query", C.c_uint32), ("rows_total", C.c_uint32), ("max_words", C.c_uint32), ("max_q_words", C.c_uint32),
                ("dim_dict", C.c_uint32), ("dim_input", C.c_uint32), ("max_line", C.c_uint32), ("dim_word", C.c_uint32),
                ("row_off", C.POINTER(C.c_uint32)), ("story_words", C.POINTER(C.c_uint16)),
                ("question_words", C.POINTER(C.c_uint16)), ("answer", C.POINTER(C.c_uint32))]


_proto("qmann_dataset_load", C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Dataset)])
_proto("qmann_dataset_free", None, [C.POINTER(Dataset)])


def load_dataset(train_path, test_path, max_sen_len=50, n_train_cap=0, n_test_cap=0):
    """qmann_dataset_load -> dict of numpy arrays (copies; the C arrays are released)."""
    import numpy as np
    ds = Dataset()
    check(lib.qmann_dataset_load(str(train_path).encode(), str(test_path).encode(), max_sen_len, n_train_cap, n_test_cap,
                                 C.byref(ds)), "qmann_dataset_load")
    try:
        nq, rows = ds.n_query, ds.rows_total
        out = dict(n_query=nq, rows_total=rows, dim_dict=ds.dim_dict, dim_input=ds.dim_input, max_line=ds.max_line, dim_word=ds.dim_word,
                   row_off=np.ctypeslib.as_array(ds.row_off, (nq + 1,)).copy(),
                   story_words=np.ctypeslib.as_array(ds.story_words, (max(rows, 1), ds.max_words))[:rows].copy(),
                   question_words=np.ctypeslib.as_array(ds.question_words, (max(nq, 1), ds.max_q_words))[:nq].copy(),
                   answer=np.ctypeslib.as_array(ds.answer, (max(nq, 1),))[:nq].copy())
    finally:
        lib.qmann_dataset_free(C.byref(ds))
    return out


# include/qmann_batch.h return codes
QMANN_OK, QMANN_EINVAL, QMANN_ERANGE, QMANN_EUNSUPPORTED, QMANN_EIO, QMANN_EHIP, QMANN_ECOMM = 0, -1, -2, -3, -4, -5, -6


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


def shard_range(n_query: int, rank: int, world: int):
    """qmann_shard_range: the contiguous [lo, hi) of `rank`."""
    lo, hi = C.c_uint32(), C.c_uint32()
    lib.qmann_shard_range(n_query, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value
