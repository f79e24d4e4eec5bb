"""Runs ON THE GPU BOX: per-hop fixed cost of a streaming hop kernel from a fit time = a + b.S
(same total number of rows at every S).  usage: scan_intercept.py [mode] [D] [nb]"""
import sys, time, numpy as np, torch
sys.path.insert(0, "tests")
from conftest import load_pkg
load_pkg()
import qmann_amd.model as model
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 10
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
cfg = model.babi_cfg(256, mode, 0, iwl=5, D=D, en_mq=False); cfg["num_bit"] = nb
rng = np.random.default_rng(0)
wts = {"w_h": [rng.normal(0, 1, (D, D)).astype(np.float32) for _ in range(3)], "w_ans": rng.normal(0, .1, (256, D)).astype(np.float32)}
net = model.QNet(cfg, wts)
total_rows = 4096 * 10000 // (2 if mode == 1 else 1)
res = []
for S in (2500, 5000, 10000, 20000):
    B = total_rows // S
    g = torch.Generator(device=dev); g.manual_seed(1)
    keys = torch.randint(-128, 128, (3, B * S, net.Dp), dtype=torch.int8, device=dev, generator=g)   # any byte is a valid sign-magnitude code
    vals = keys
    u0 = (torch.randn((B, D), device=dev, generator=g) * 30).round_().clamp_(-127, 127) / 4.0
    ro = (torch.arange(B + 1, device=dev, dtype=torch.int64) * S).to(torch.int32)
    planes = net.pack_planes(keys, nb) if mode in (10, 11) else None
    run = (lambda: net.hops_packed(planes, vals, ro, S, u0)) if planes is not None else (lambda: net.hops(keys, vals, ro, S, u0))
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    res.append((S, B, ms))
    print(f"S={S:6d} B={B:6d}  {ms:.3f} ms   per query-hop {ms * 1e3 / (3 * B):.2f} us", flush=True)
    del keys, vals, planes
# time = n_queryhops * (a + b S) / concurrency  -> per query-hop cost in units of the S=10000 scan
(S1, B1, t1), (S2, B2, t2) = res[0], res[-1]
# t = B*(a + b*S)*c ; B*S constant = R  ->  t = c*(a*R/S + b*R)
a_over_b = (t1 - t2) / (1 / S1 - 1 / S2) / ((t2 - (t1 - t2) / (1 / S1 - 1 / S2) / S2)) 
print(f"fixed cost per hop = {a_over_b:.0f} slot-equivalents ({a_over_b / 10000 * 100:.1f} % of a 10 000-slot scan)")
