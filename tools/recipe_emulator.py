#!/usr/bin/env python
"""CPU emulation of storage / arithmetic recipes of the train step's FORWARD pass on the float64 oracle (developer tool; no GPU).

Which operands may be rounded to what before north_star's 1e-3 on the accent / disc probabilities and the CTC loss is lost?  The oracle
graph (oracle/model_torch.py) runs in float64 with selected operands passed through a rounding function first; everything else stays
exact, so each row isolates what ONE choice costs.  Knobs (each takes a format):

    conv_x   the activated operand of every trunk conv (what the patch fill hands the MFMA)
    conv_w   the conv kernels
    conv_y   what a conv leaves in HBM (storage of the residual stream / BatchNorm inputs)
    gemm_x   operands of the Dense / GRU-projection GEMMs (rounded at the LayerNorm outputs and at the trunk output)
    gemm_w   Dense kernels and GRU input kernels
    gru_u    GRU recurrent kernels          gru_h   the hidden state as the recurrence's matrix operand

formats: exact | fp16 | bf16 | bf16x2 (hi + lo bf16 planes: the operand of a three-MFMA product; 16 mantissa bits) | fp16x2 | fp32

    python tools/recipe_emulator.py                      # the table of DESIGN.md (cfg3 parity size, max over 3 seeds)
    python tools/recipe_emulator.py conv_x=fp16 conv_w=fp16 conv_y=fp16      # one recipe
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import model_torch as MT                                  # noqa: E402  (developer tool: the oracle is the reference here)

CASES = {
    "cfg2": dict(input_shape=[200, 80, 1], res_type="res34", res_filters=32, hidden_dim=64, mto="gvlad", vlad_clusters=8, ghost_clusters=2,
                 disc_enable=True, metric_loss="arcface", B=8),
    "cfg3": dict(input_shape=[200, 80, 1], res_type="res34", res_filters=32, hidden_dim=64, mto="gvlad", vlad_clusters=8, ghost_clusters=2,
                 disc_enable=True, ctc_enable=True, bpe_classes=50, max_ctc_len=8, metric_loss="circleloss", B=8),
    "cfg3_h256": dict(input_shape=[200, 80, 1], res_type="res34", res_filters=32, hidden_dim=256, mto="gvlad", vlad_clusters=8, ghost_clusters=2,
                      disc_enable=True, ctc_enable=True, bpe_classes=50, max_ctc_len=8, metric_loss="circleloss", B=8),
}
KNOBS = ("conv_x", "conv_w", "conv_y", "gemm_x", "gemm_w", "gru_u", "gru_h")


def _round(x, fmt):
    if fmt == "exact":
        return x
    if fmt == "fp32":
        return x.float().double()
    if fmt in ("fp16", "bf16"):
        return x.to(torch.float16 if fmt == "fp16" else torch.bfloat16).double()
    if fmt in ("fp16x2", "bf16x2"):
        t = torch.float16 if fmt == "fp16x2" else torch.bfloat16
        hi = x.to(t).double()
        return hi + (x - hi).to(t).double()
    raise ValueError(fmt)


def q(x, fmt):
    """Rounded value forward, identity backward."""
    if fmt == "exact":
        return x
    return x + (_round(x.detach(), fmt) - x.detach())


class Recipe:
    def __init__(self, **kw):
        for k in KNOBS:
            setattr(self, k, kw.pop(k, "exact"))
        if kw:
            raise ValueError("unknown knobs %s" % list(kw))

    def __str__(self):
        s = " ".join("%s=%s" % (k, getattr(self, k)) for k in KNOBS if getattr(self, k) != "exact")
        return s or "exact"


def run(case, recipe, seed):
    kw = dict(CASES[case]); B = kw.pop("B")
    cfg = MT.Cfg(**kw)
    net = MT.TorchSARNet(cfg, seed=seed)
    batch = MT.synthetic_batch(cfg, B, seed=77 + seed)
    with torch.no_grad():
        want = MT.forward(net.P, batch, cfg, state=None, training=True)
    o_conv, o_ln, o_gru = MT.conv2d, MT.ln, MT.gru_dir

    def conv2d(x, w, b, stride, padding):
        first = x.shape[-1] == 1                                      # the stem reads the fp32 input: its operand is what the feeder wrote
        y = o_conv(q(x, "fp32" if first and recipe.conv_x == "exact" else recipe.conv_x), q(w, recipe.conv_w), b, stride, padding)
        return q(y, recipe.conv_y)

    def ln(x, g, b):
        return q(o_ln(x, g, b), recipe.gemm_x)

    def gru_dir(x, k, rk, bias, reverse):
        Bn, L, _ = x.shape; H = rk.shape[0]
        xa = x @ q(k, recipe.gemm_w) + bias[:3 * H]
        rkq = q(rk, recipe.gru_u)
        h = x.new_zeros(Bn, H); outs = [None] * L
        for t in (range(L - 1, -1, -1) if reverse else range(L)):
            ha = q(h, recipe.gru_h) @ rkq + bias[3 * H:]
            z = torch.sigmoid(xa[:, t, :H] + ha[:, :H]); r = torch.sigmoid(xa[:, t, H:2 * H] + ha[:, H:2 * H])
            hh = torch.tanh(xa[:, t, 2 * H:] + r * ha[:, 2 * H:])
            h = z * h + (1 - z) * hh
            outs[t] = h
        return torch.stack(outs, 1)

    P = dict(net.P)
    for n in P:
        if n.endswith("/kernel") and not n.startswith("conv2d") and "recurrent" not in n and "/forward/" not in n and "/backward/" not in n:
            P[n] = q(P[n], recipe.gemm_w)
    MT.conv2d, MT.ln, MT.gru_dir = conv2d, ln, gru_dir
    try:
        with torch.no_grad():
            # the trunk output feeds CNN_LIN: a GEMM operand as well
            got = MT.forward(P, batch, cfg, state=None, training=True)
    finally:
        MT.conv2d, MT.ln, MT.gru_dir = o_conv, o_ln, o_gru
    row = {}
    for k in ("y_accent", "y_disc"):
        if k in want:
            row[k] = float((got[k] - want[k]).abs().max())
    if "y_ctc_loss" in want:
        row["ctc_rel"] = float((got["y_ctc_loss"] - want["y_ctc_loss"]).abs().max() / want["y_ctc_loss"].abs().max())
    row["cnn_rel"] = float((got["cnn"] - want["cnn"]).abs().max() / want["cnn"].abs().max())
    return row


TABLE = [
    Recipe(),
    Recipe(conv_x="fp16", conv_w="fp16", conv_y="fp16"),
    Recipe(conv_x="bf16", conv_w="bf16", conv_y="bf16"),
    Recipe(conv_x="fp16", conv_w="fp16"),
    Recipe(conv_x="fp16"),
    Recipe(conv_w="fp16"),
    Recipe(conv_y="fp16"),
    Recipe(conv_x="bf16x2", conv_w="bf16x2", conv_y="bf16x2"),
    Recipe(conv_x="bf16x2", conv_w="bf16x2", conv_y="bf16x2", gemm_x="bf16x2", gemm_w="bf16x2", gru_u="bf16x2", gru_h="bf16x2"),
    Recipe(gemm_x="bf16", gemm_w="bf16", gru_u="bf16", gru_h="bf16"),
    Recipe(gemm_x="fp16", gemm_w="fp16", gru_u="fp16", gru_h="fp16"),
    Recipe(gemm_x="fp16", gemm_w="fp16"),
    Recipe(gru_u="fp16", gru_h="fp16"),
    Recipe(gru_u="fp16"),
    Recipe(gru_u="bf16"),
    Recipe(gru_h="fp16"),
    Recipe(gemm_w="fp16"),
    Recipe(gemm_x="fp16"),
    Recipe(conv_x="bf16x2", conv_w="bf16x2", conv_y="bf16x2", gemm_x="bf16x2", gemm_w="bf16x2", gru_u="fp16", gru_h="fp16x2"),
]


def main():
    args = [a for a in sys.argv[1:] if "=" in a]
    cases = [a for a in sys.argv[1:] if "=" not in a] or ["cfg3", "cfg2"]
    recipes = [Recipe(**dict(a.split("=") for a in args))] if args else TABLE
    seeds = range(int(os.environ.get("RE_SEEDS", "3")))
    for case in cases:
        print("== %s (B = %d, %d frames; max over %d seeds)" % (case, CASES[case]["B"], CASES[case]["input_shape"][0], len(seeds)))
        print("%-110s %10s %10s %10s %10s" % ("recipe", "accent", "disc", "ctc_rel", "cnn_rel"))
        for r in recipes:
            rows = [run(case, r, s) for s in seeds]
            mx = {k: max(row.get(k, 0.0) for row in rows) for k in ("y_accent", "y_disc", "ctc_rel", "cnn_rel")}
            print("%-110s %10.2e %10.2e %10.2e %10.2e" % (str(r), mx["y_accent"], mx["y_disc"], mx["ctc_rel"], mx["cnn_rel"]))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
