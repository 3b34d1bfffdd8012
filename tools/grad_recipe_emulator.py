#!/usr/bin/env python
"""CPU emulation of BACKWARD-pass arithmetic recipes of the conv trunk on the float64 oracle (developer tool; no GPU).  Companion of
tools/recipe_emulator.py (forward recipes).  The forward pass stays exact; in the backward pass of every trunk conv selected operands are rounded:

    dgrad_g   the gradient operand of the input-gradient conv (what the gradient fill hands the MFMA) and of the weight gradient
    dgrad_w   the kernel as the input-gradient conv's operand
    store_g   what an input-gradient conv leaves in HBM (the gradient of the residual stream)
    wgrad_x   the activation operand of the weight gradient

formats as in recipe_emulator.py; `scale` = the static gradient scale the 2-byte formats carry (64 x batch).  Reported: flat parameter gradient
relative L2 error, the worst per-tensor cosine and max-abs-relative error against the exact backward pass.
    python tools/grad_recipe_emulator.py                                  # table
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import model_torch as MT                                  # noqa: E402
from recipe_emulator import CASES, _round                             # noqa: E402


def rq(x, fmt, scale=1.0):
    if fmt == "exact":
        return x
    return _round(x * scale, fmt) / scale


def make_conv(rec, scale):
    class Conv(torch.autograd.Function):
        @staticmethod
        def forward(ctx, xc, wc, b, stride):
            ctx.save_for_backward(xc, wc); ctx.stride = stride; ctx.has_b = b is not None
            return F.conv2d(xc, wc, b, stride=stride)

        @staticmethod
        def backward(ctx, gy):
            xc, wc = ctx.saved_tensors
            gq = rq(gy, rec["dgrad_g"], scale)
            gx = torch.nn.grad.conv2d_input(xc.shape, rq(wc, rec["dgrad_w"]), gq, stride=ctx.stride)
            gx = rq(gx, rec["store_g"], scale)
            gw = torch.nn.grad.conv2d_weight(rq(xc, rec["wgrad_x"]), wc.shape, gq, stride=ctx.stride)
            gb = gy.sum((0, 2, 3)) if ctx.has_b else None
            return gx, gw, gb, None
    return Conv


def make_mm(rec, scale):
    """seq_mm whose backward rounds: gemm_g the gradient operand (scaled) of both products, gemm_w the kernel in dX = dY W^T, gemm_x the
    activation in dW = X^T dY."""
    class MM(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w

        @staticmethod
        def backward(ctx, gy):
            x, w = ctx.saved_tensors
            gq = rq(gy, rec.get("gemm_g", "exact"), scale)
            gx = gq @ rq(w, rec.get("gemm_w", "exact")).T
            x2, g2 = rq(x, rec.get("gemm_x", "exact")).reshape(-1, x.shape[-1]), gq.reshape(-1, gq.shape[-1])
            return gx, x2.T @ g2
    return MM


def run(case, rec, seed):
    kw = dict(CASES[case]); B = kw.pop("B")
    cfg = MT.Cfg(**kw)
    net = MT.TorchSARNet(cfg, seed=seed)
    batch = MT.synthetic_batch(cfg, B, seed=77 + seed)
    _, _, _, G0 = net.forward_backward(batch)
    o_conv = MT.conv2d
    Conv = make_conv(rec, 2.0 ** int(round(np.log2(64.0 * B))))

    def conv2d(x, w, b, stride, padding):
        kh, kw_ = w.shape[0], w.shape[1]
        xc = x.permute(0, 3, 1, 2)
        if padding == "same":
            pt, pb = MT._same(x.shape[1], kh, stride); pl, pr = MT._same(x.shape[2], kw_, stride)
            xc = F.pad(xc, (pl, pr, pt, pb))
        return Conv.apply(xc, w.permute(3, 2, 0, 1), b, stride).permute(0, 2, 3, 1)
    MT.conv2d = conv2d
    o_mm = MT.seq_mm
    MM = make_mm(rec, 2.0 ** int(round(np.log2(64.0 * B))))
    MT.seq_mm = lambda x, w: MM.apply(x, w)
    try:
        _, _, _, G1 = net.forward_backward(batch)
    finally:
        MT.conv2d = o_conv; MT.seq_mm = o_mm
    names = list(G0)
    f0 = torch.cat([G0[k].ravel() for k in names]); f1 = torch.cat([G1[k].ravel() for k in names])
    l2 = float((f1 - f0).norm() / f0.norm())
    big = max(float(G0[k].norm()) for k in names)
    worst_cos, worst_rel = 1.0, 0.0
    for k in names:
        a, b = G0[k].ravel(), G1[k].ravel()
        if float(a.norm()) < 1e-3 * big:
            continue
        worst_cos = min(worst_cos, float(a @ b / (a.norm() * b.norm())))
        worst_rel = max(worst_rel, float((a - b).abs().max() / a.abs().max()))
    return l2, worst_cos, worst_rel


TABLE = [
    dict(dgrad_g="exact", dgrad_w="exact", store_g="exact", wgrad_x="exact"),
    dict(dgrad_g="fp16", dgrad_w="exact", store_g="exact", wgrad_x="fp16"),          # what ships at r4: x3 input gradients, fp16 weight-gradient operands (approx.: the x3 dgrad reads the unrounded operand)
    dict(dgrad_g="fp16", dgrad_w="fp16", store_g="fp16", wgrad_x="fp16"),            # the whole backward chain single-pass fp16
    dict(dgrad_g="fp16", dgrad_w="fp16", store_g="fp32", wgrad_x="fp16"),            # single-pass fp16 products, fp32 residual-stream gradient in HBM
    dict(dgrad_g="bf16", dgrad_w="bf16", store_g="bf16", wgrad_x="bf16"),
    # + the per-frame Dense / GRU-projection GEMMs' backward single-pass (round 5, second half): gradient operand fp16 under the scale
    dict(dgrad_g="fp16", dgrad_w="fp16", store_g="fp16", wgrad_x="fp16", gemm_g="fp16", gemm_w="fp16", gemm_x="fp16"),
    dict(dgrad_g="fp16", dgrad_w="fp16", store_g="fp16", wgrad_x="fp16", gemm_g="fp16", gemm_w="exact", gemm_x="fp16"),
    dict(dgrad_g="fp16", dgrad_w="fp16", store_g="fp16", wgrad_x="fp16", gemm_g="bf16", gemm_w="bf16", gemm_x="bf16"),
    dict(dgrad_g="exact", dgrad_w="exact", store_g="exact", wgrad_x="exact", gemm_g="fp16", gemm_w="fp16", gemm_x="fp16"),
]

if __name__ == "__main__":
    torch.set_num_threads(8)
    cases = [a for a in sys.argv[1:]] or ["cfg3", "cfg2"]
    seeds = range(int(os.environ.get("RE_SEEDS", "3")))
    for case in cases:
        print("== %s (max over %d seeds): flat gradient rel. L2 | worst tensor cosine | worst tensor max-abs rel." % (case, len(seeds)))
        for rec in TABLE:
            rows = [run(case, rec, s) for s in seeds]
            print("%-90s %10.2e %10.6f %10.2e" % (" ".join("%s=%s" % kv for kv in rec.items()), max(r[0] for r in rows), min(r[1] for r in rows), max(r[2] for r in rows)))
