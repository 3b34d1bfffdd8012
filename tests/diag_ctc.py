"""Diagnostic (not collected by pytest; lives under tests/ because only tests may import oracle/): CTC kernel vs oracle on one batch, including uninitialised workspace."""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ops_np as O
from aesrc2020_amd import _lib as L
# churn the allocator with garbage first
junk = [torch.full((n,), 1e36, device="cuda") for n in (1000, 10000, 100000, 1000000, 5000000)]
junk2 = [torch.full((n,), float("nan"), device="cuda") for n in (3000, 30000, 300000, 3000000)]
del junk, junk2
def dev(a, dt=torch.float32): return torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
B, Lq, V, maxU, Umax = 4, 12, 9, 8, 4
rng = np.random.default_rng(V)
p = O.softmax(rng.standard_normal((B, Lq, V)) * 2)
lab = np.full((B, maxU), 2.0, np.float32); ll = np.zeros((B, 1), np.int32)
for b in range(B):
    u = int(rng.integers(1, Umax + 1)); ll[b] = u; lab[b, :u] = rng.integers(0, V - 1, u)
il = np.full((B, 1), Lq, np.int32); il[1] = Lq - 3
loss_ref, grad_ref = O.ctc_batch_cost_fwd_bwd(p, lab, il, ll)
pd, labd, ild, lld = dev(p), dev(lab), dev(il.ravel(), torch.int32), dev(ll.ravel(), torch.int32)
wsb = L.ctc_workspace_bytes(B, Lq, maxU)
for trial, wsinit in enumerate(("zeros", "garbage")):
    ws = torch.zeros(wsb // 4, device="cuda") if wsinit == "zeros" else torch.full((wsb // 4,), 1e36, device="cuda")
    loss = torch.zeros(B, device="cuda")
    L.call("sar_ctc_batch_cost_fwd_f32", pd, labd, ild, lld, loss, ws, wsb, B, Lq, V, maxU)
    dp = torch.full((B, Lq, V), 7.0, device="cuda")
    L.call("sar_ctc_batch_cost_bwd_f32", pd, labd, ild, lld, loss, dev(np.ones(B)), ws, wsb, dp, B, Lq, V, maxU)
    d = np.abs(dp.cpu().numpy() - grad_ref)
    idx = np.unravel_index(d.argmax(), d.shape)
    print(wsinit, "loss", loss.cpu().numpy(), loss_ref.ravel(), "max err", d.max(), "at", idx, "lens", il.ravel(), ll.ravel())
