"""Per-kernel timing of the recompute-form stem (csrc/stem_bf16.hip) at a bench shape: python tools/stem_bench.py [B T D C0]."""
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aesrc2020_amd import _lib as L  # noqa: E402
from oracle.ops_np import same_pads  # noqa: E402  (geometry helper only)

B, T, D, C0 = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 400, 80, 64)
Ho, pt, _ = same_pads(T, 7, 2); Wo, pl, _ = same_pads(D, 7, 2)
Hp, ppt, _ = same_pads(Ho, 3, 2); Wp, ppl, _ = same_pads(Wo, 3, 2)
rng = np.random.default_rng(0)
x = torch.tensor(rng.uniform(0, 1, (B, T, D)), dtype=torch.float32, device="cuda")
w = torch.tensor(rng.standard_normal((7, 7, 1, C0)) * 0.2, dtype=torch.float32, device="cuda")
b = torch.zeros(C0, device="cuda"); gamma = torch.ones(C0, device="cuda"); beta = torch.zeros(C0, device="cuda")
NB = L.bn_bins()
bins = torch.zeros(NB, 2, C0, device="cuda"); bbins = torch.zeros(NB, 2, C0, device="cuda"); saved = torch.zeros(4, C0, device="cuda")
mm = torch.zeros(C0, device="cuda"); mv = torch.ones(C0, device="cuda"); dg = torch.zeros(C0, device="cuda"); db = torch.zeros(C0, device="cuda")
pooled = torch.zeros(B, Hp, Wp, C0, dtype=torch.bfloat16, device="cuda"); idx = torch.zeros(B, Hp, Wp, C0, dtype=torch.uint8, device="cuda")
g = torch.randn(B, Hp, Wp, C0, device="cuda").to(torch.bfloat16)
dw = torch.zeros(7, 7, 1, C0, device="cuda"); work = torch.empty(L.stem16_wgrad_workspace_bytes(C0) // 4, device="cuda")
R = B * Ho * Wo


def ref(stats):
    r = L.BnRef()
    r.gamma, r.beta, r.saved = gamma.data_ptr(), beta.data_ptr(), saved.data_ptr()
    r.bins = bins.data_ptr() if stats == "fwd" else None
    r.bbins = bbins.data_ptr() if stats == "bwd" else None
    r.moving_mean, r.moving_var, r.dgamma, r.dbeta = mm.data_ptr(), mv.data_ptr(), dg.data_ptr(), db.data_ptr()
    r.inv_n, r.unbias, r.eps, r.momentum, r.write = 1.0 / R, R / (R - 1.0), 1e-3, 0.99, 1
    return r


conv = (B, T, D, Ho, Wo, C0, pt, pl); pool = (Hp, Wp, ppt, ppl)
steps = {
    "stats": lambda: L.call("sar_stem16_stats", x, w, b, bins, *conv, 0),
    "pool_fwd": lambda: L.call("sar_stem16_pool_fwd", x, w, b, ref("fwd"), pooled, idx, *conv, *pool),
    "bwd_reduce": lambda: L.call("sar_stem16_bwd_reduce", g, pooled, idx, x, w, b, ref("bwd"), work, *conv, *pool),
    "wgrad(+wsum)": lambda: L.call("sar_stem16_wgrad", x, w, b, g, pooled, idx, ref("bwd"), dw, work, *conv, *pool),
}
for name, fn in steps.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("%-14s %8.2f us" % (name, e0.elapsed_time(e1) * 1000 / n))
