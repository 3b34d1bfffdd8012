#!/usr/bin/env python
"""Developer tool: random conv shapes, bf16x3 kernels vs the exact-fp32 kernels (fwd / dgrad / wgrad), max relative deviation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aesrc2020_amd import _lib as L
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
worst = []
for it in range(int(os.environ.get("N", 60))):
    ci = int(rng.choice([16, 32, 64, 128])); co = int(rng.choice([16, 32, 64, 128]))
    k, s = [(3, 1), (3, 2), (1, 1), (1, 2)][int(rng.integers(0, 4))]
    B = int(rng.integers(1, 6)); H = int(rng.integers(3, 40)); W = int(rng.integers(3, 24))
    if k == 3:
        Ho, Wo = -(-H // s), -(-W // s); pt = max((Ho - 1) * s + k - H, 0) // 2; pl = max((Wo - 1) * s + k - W, 0) // 2
    else:
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1; pt = pl = 0
    x = torch.randn(B, H, W, ci, device="cuda"); w = torch.randn(k, k, ci, co, device="cuda") * 0.1; b = torch.randn(co, device="cuda")
    pro = bool(rng.integers(0, 2)); res = bool(rng.integers(0, 2)); acc = int(rng.integers(0, 2))
    sc = torch.rand(ci, device="cuda") + 0.5 if pro else None; sh = torch.randn(ci, device="cuda") * 0.3 if pro else None
    r = torch.randn(B, Ho, Wo, co, device="cuda") if res else None
    dy = torch.randn(B, Ho, Wo, co, device="cuda")
    out = {}
    for mode in ("fp32", "bf16x3"):
        L.set_conv_math(mode)
        y = torch.zeros(B, Ho, Wo, co, device="cuda"); dx = torch.ones(B, H, W, ci, device="cuda"); dw = torch.zeros_like(w); db = torch.zeros(co, device="cuda")
        L.call("sar_conv2d_fwd_f32", x, w, b, sc, sh, r, y, B, H, W, ci, Ho, Wo, co, k, k, s, pt, pl)
        L.call("sar_conv2d_dgrad_f32", dy, w, dx, B, H, W, ci, Ho, Wo, co, k, k, s, pt, pl, acc)
        L.call("sar_conv2d_wgrad_f32", x, dy, sc, sh, dw, db, B, H, W, ci, Ho, Wo, co, k, k, s, pt, pl)
        torch.cuda.synchronize()
        out[mode] = (y, dx, dw, db)
    for name, a, c in zip(("fwd", "dgrad", "wgrad", "dbias"), out["fp32"], out["bf16x3"]):
        e = float((a - c).abs().max() / (a.abs().max() + 1e-30))
        worst.append((e, name, (B, H, W, ci, co, k, s, pro, res, acc)))
worst.sort(reverse=True)
for e, name, shp in worst[:12]:
    print("%.2e %-6s %s" % (e, name, shp))
