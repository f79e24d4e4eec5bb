import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import bench
bench.load_pkg()
import qmann_amd.model as model, qmann_amd.abi as abi
from pyoracle import Oracle
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["synth10k_d256_ham"]
S, D, V, mode, nb = wl["S"], wl["D"], wl["V"], wl["mode"], wl["nb"]
B, H = 24, 3
cfg = model.babi_cfg(V, attention_mode=mode, softmax_base=0, iwl=5, n_hop=H, D=D, en_mq=False); cfg["num_bit"] = nb
wts = bench.make_params(cfg, D, V, seed=0x51A44)
wts["w_ans"] = (np.clip(np.rint(wts["w_ans"] * 64.0 * 4), -127, 127) / 64.0).astype(np.float32)
net = model.QNet(cfg, wts)
gen = torch.Generator(device=dev); gen.manual_seed(0x51A44)
keys = bench.gauss_i8((H, B * S, 256), wl["sk"], gen, dev); vals = bench.gauss_i8((H, B * S, 256), wl["sv"], gen, dev)
u0 = (torch.randn((B, D), device=dev, generator=gen) * wl["su"]).round_().clamp_(-127, 127) / 4.0
row_off = (torch.arange(B + 1, device=dev, dtype=torch.int64) * S).to(torch.int32)
planes = net.pack_planes(keys, nb)
u, taps = net.hops_packed(planes, vals, row_off, S, u0, taps=True)
w_i8 = net.quantize_i8(net.w_ans, (1, 6), abi.CODE_TWOS)
pred_i, probs_i, _, _, logits = net.answer_i8(u, w_i8, (1, 6), want_probs=True)
pred_f, probs_f, _, _ = net.answer(u, want_probs=True)
torch.cuda.synchronize()
ora = Oracle(); m = ora.make_model(cfg, wts)
for q in range(B):
    kf = np.stack([model.from_signmag(keys[h, q*S:(q+1)*S, :D].cpu().numpy()).astype(np.float32) / 4 for h in range(H)])
    vf = np.stack([model.from_signmag(vals[h, q*S:(q+1)*S, :D].cpu().numpy()).astype(np.float32) / 4 for h in range(H)])
    op, t = ora.forward_mem(m, kf, vf, u0[q].cpu().numpy())
    gu = taps.u[q].cpu().numpy()
    same_u = [bool(np.array_equal(gu[h], t["u"][h])) for h in range(H)]
    same_sc = [bool(np.array_equal(taps.scores[h, q*S:(q+1)*S].cpu().numpy(), t["scores"][h])) for h in range(H)]
    pr = [float(np.abs(taps.probs[h, q*S:(q+1)*S].cpu().numpy() - t["probs"][h]).max()) for h in range(H)]
    top = np.sort(t["probs"][0])[-4:]
    print(q, "pred gpu_i8", int(pred_i[q]), "gpu_f32", int(pred_f[q]), "oracle", op, "u ok", same_u, "scores ok", same_sc, "dprob", pr, "top p hop0", top, "logit diff", float(np.abs(logits[q].cpu().numpy() - t["logits"]).max()))
