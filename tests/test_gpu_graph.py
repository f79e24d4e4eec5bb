"""The batched forward inside a HIP graph.  Every entry point of the library takes a stream and returns without synchronising
(include/qmann_model.h); once a model's workspace has its size (first call), a forward is a fixed sequence of kernel launches
and memsets on that stream -- no allocation, no host read-back -- so a host that serves SMALL batches can capture it once
(hipStreamBeginCapture .. EndCapture) and replay the graph per batch: one graph launch instead of 6-9 kernel launches.  The
reference has no counterpart (one query at a time, 31 launches each: MemN2N/MemN2N.c:2626-2697).
torch is used for the capture API only (torch.cuda.CUDAGraph = hipGraph on ROCm); the captured work is the library's."""
import numpy as np
import pytest

from conftest import load_pkg

pytestmark = pytest.mark.gpu


def _words(rng, n, V, dd, max_words, with_time, n_sen=None):
    out = np.full((n, max_words), 0xFFFF, np.uint16)
    for r in range(n):
        k = int(rng.integers(1, max_words - 1))
        out[r, :k] = rng.integers(1, dd, k)
        if with_time:
            out[r, k] = dd + int(rng.integers(0, V - dd))
    return out


@pytest.mark.parametrize("mode,nb,B,cap", [(2, 8, 96, 10), (3, 8, 96, 10), (11, 4, 96, 10),
                                            # a batch that is split by story length, its two hop kernels side by side on two
                                            # streams (csrc/hops_lean.h::launch_lean): the capture takes both branches
                                            (2, 8, 40000, 40), (3, 8, 40000, 40)])
def test_forward_words_replays_from_a_captured_graph(mode, nb, B, cap):
    import torch
    load_pkg()
    import qmann_amd.model as model
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5 + mode)
    V, dd, D, H = 40, 30, 60, 3
    cfg = model.babi_cfg(V, mode, 0, iwl=5, en_mq=True)
    cfg["num_bit"] = nb
    wts = {"w_q": rng.normal(0, 1.0, (D, V)).astype(np.float32), "w_ans": rng.normal(0, 0.3, (V, D)).astype(np.float32),
           "w_a": [rng.normal(0, 1.0, (D, V)).astype(np.float32) for _ in range(H)],
           "w_c": [rng.normal(0, 1.0, (D, V)).astype(np.float32) for _ in range(H)],
           "w_h": [rng.normal(0, 1.0, (D, D)).astype(np.float32) for _ in range(H)]}
    n_sen = rng.integers(1, 11, B).astype(np.int64)
    if cap > 10:
        n_sen[::9] = rng.integers(17, cap + 1, len(n_sen[::9]))                   # one story in nine is long (mean length stays < 16)
    rows = int(n_sen.sum())
    row_off = torch.from_numpy(np.concatenate([[0], np.cumsum(n_sen)]).astype(np.int32)).to(dev)

    def batch(seed):
        r = np.random.default_rng(seed)
        return (torch.from_numpy(_words(r, rows, V, dd, 8, True).view(np.int16)).to(dev),
                torch.from_numpy(_words(r, B, V, dd, 8, False).view(np.int16)).to(dev),
                torch.from_numpy(r.integers(0, V, B).astype(np.int32)).to(dev))
    s = torch.cuda.Stream(device=dev)
    hm = model.HostModel(cfg, wts, device="cuda:0", stream=s.cuda_stream)        # every launch of this model goes to stream s
    plain = model.HostModel(cfg, wts, device="cuda:0")                           # the same model on the default stream: the check
    sw, qw, ans = batch(1)                                                       # static input buffers of the graph
    with torch.cuda.stream(s):
        hm.forward_words(sw, qw, row_off, cap, ans)                              # warm-up: the workspace takes its size
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        pred, cost, match = hm.forward_words(sw, qw, row_off, cap, ans)          # captured, not run
    for seed in (2, 3, 4):
        nsw, nqw, nans = batch(seed)
        sw.copy_(nsw); qw.copy_(nqw); ans.copy_(nans)                            # new batch into the graph's input buffers
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        p2, c2, m2 = plain.forward_words(nsw, nqw, row_off, cap, nans)
        torch.cuda.synchronize()
        # (cost / match are single words the forward ACCUMULATES into: the captured memset of _out zeroes them per replay? no --
        # they are zeroed by torch.zeros inside the capture, which is part of the graph)
        assert torch.equal(pred, p2), seed
        assert int(match.item()) == int(m2.item()) and float(cost.item()) == pytest.approx(float(c2.item()), rel=1e-5)
    assert len({int(x) for x in pred.cpu().numpy()}) > 1                         # (not a degenerate constant prediction)
