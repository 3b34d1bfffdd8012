#!/usr/bin/env python
"""What each storage / arithmetic recipe of the train step costs in accuracy: deviation from the float64 oracle
(oracle/model_torch.py) of the accent / disc probabilities, the per-utterance CTC loss and the parameter gradients, for

    f32      fp32 storage, fp32 MFMA                       bf16x3   fp32 storage, split-precision convs (3 bf16 MFMAs per product)
    bf16     bf16 trunk + bf16-MFMA Dense / GRU GEMMs       fp16     fp16 trunk (static gradient scale) + bf16-MFMA GEMMs
    bf16+g32 bf16 trunk, fp32 GEMMs                         fp16+g32 fp16 trunk, fp32 GEMMs

on the parity configurations of tests/test_gpu_bf16_model.py (B = 8, 200 frames).  For the fp16 trunk it also prints the range the
scaled activation gradients occupy (largest magnitude against fp16's 65504; share of non-zero values below the smallest normal 6.1e-5).

    python tools/numerics_table.py [case ...]          SAR_F16_GSCALE=<S> to try another gradient scale
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
from oracle import model_torch as MT                                  # noqa: E402  (developer tool: the oracle is the reference here)
from aesrc2020_amd import _lib                                         # noqa: E402
from aesrc2020_amd.model import SAR_Net                                # noqa: E402

CASES = {
    "cfg2_gvlad_arcface": dict(input_shape=[200, 80, 1], res_type="res34", res_filters=32, hidden_dim=64, mto="gvlad",
                               vlad_clusters=8, ghost_clusters=2, disc_enable=True, metric_loss="arcface", B=8),
    "cfg3_ctc_circle": dict(input_shape=[200, 80, 1], res_type="res34", res_filters=32, hidden_dim=64, mto="gvlad",
                            vlad_clusters=8, ghost_clusters=2, disc_enable=True, ctc_enable=True, bpe_classes=50,
                            max_ctc_len=8, metric_loss="circleloss", B=8),
    "cfg3_h256": dict(input_shape=[200, 80, 1], res_type="res34", res_filters=32, hidden_dim=256, mto="gvlad",
                      vlad_clusters=8, ghost_clusters=2, disc_enable=True, ctc_enable=True, bpe_classes=50,
                      max_ctc_len=8, metric_loss="circleloss", B=8),
    "smoke_b4_t100": dict(input_shape=[100, 80, 1], res_type="res34", res_filters=32, hidden_dim=256, mto="gvlad", vlad_clusters=8,
                          ghost_clusters=2, disc_enable=True, ctc_enable=True, bpe_classes=50, max_ctc_len=8, metric_loss="circleloss", B=4),
    "b8_t100": dict(input_shape=[100, 80, 1], res_type="res34", res_filters=32, hidden_dim=256, mto="gvlad", vlad_clusters=8,
                    ghost_clusters=2, disc_enable=True, ctc_enable=True, bpe_classes=50, max_ctc_len=8, metric_loss="circleloss", B=8),
    "b4_t200": dict(input_shape=[200, 80, 1], res_type="res34", res_filters=32, hidden_dim=256, mto="gvlad", vlad_clusters=8,
                    ghost_clusters=2, disc_enable=True, ctc_enable=True, bpe_classes=50, max_ctc_len=8, metric_loss="circleloss", B=4),
}
MODES = {"x3": ("bf16x3", "bf16x3", "bf16x3"), "x3+g32": ("bf16x3", "bf16x3", "fp32"), "f32": ("fp32", "fp32", "fp32"), "bf16x3": ("fp32", "bf16x3", "fp32"), "bf16": ("bf16", "bf16x3", "bf16"), "fp16": ("fp16", "bf16x3", "bf16"),
         "bf16+g32": ("bf16", "bf16x3", "fp32"), "fp16+g32": ("fp16", "bf16x3", "fp32")}


def cosine(a, b):
    a, b = a.ravel().astype(np.float64), b.ravel().astype(np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def run(case, mode, seed=0):
    kw = dict(CASES[case]); B = kw.pop("B")
    cfg = MT.Cfg(**kw)
    ref = MT.TorchSARNet(cfg, seed=seed)
    trunk, conv, gemm = MODES[mode]
    _lib.set_trunk_dtype(trunk); _lib.set_conv_math(conv); _lib.set_gemm_math(gemm)
    try:
        model, _ = SAR_Net(seed=seed + 1, **kw)
        model.set_weights_dict({k: v.numpy() for k, v in ref.P.items()})
        batch = MT.synthetic_batch(cfg, B, seed=77)
        out, total, parts, G = ref.forward_backward(batch)
        model.forward_backward_only(batch, batch)
        torch.cuda.synchronize()
        if os.environ.get("NT_ACT") and getattr(model.trunk, "bf16", False):   # largest stored activation of the 2-byte trunk
            mx = max(float(t.float().abs().max()) for key, t in model.trunk.store.bufs.d.items() if t.dtype == model.trunk.t16 and "wf16" not in key and "wd16" not in key)
            print("          largest stored 2-byte value of the trunk: %.4g" % mx)
    finally:
        _lib.set_trunk_dtype("fp32"); _lib.set_conv_math("bf16x3"); _lib.set_gemm_math("fp32")
    row = {}
    for k in ("y_accent", "y_disc"):
        if k in out and k in model.out:
            row[k] = float(np.abs(model.out[k].cpu().numpy() - out[k].detach().numpy()).max())
    if cfg.ctc_enable:
        got = model.out["y_ctc_loss"].cpu().numpy(); want = out["y_ctc_loss"].detach().numpy()[:, 0]
        row["ctc_rel"] = float(np.abs(got - want).max() / np.abs(want).max())
    ref_g = {r.name: G[r.name].numpy() - (2 * MT.L2 * ref.P[r.name].numpy() if r.name in ref.reg else 0.0) for r in model.store.refs}
    norms = {k: np.linalg.norm(v) for k, v in ref_g.items()}
    big = max(norms.values())
    cos_conv, cos_bn, cos_rest, rel_all = [], [], [], []
    for r in model.store.refs:
        if norms[r.name] < 1e-3 * big:
            continue
        g = r.g.cpu().numpy()
        c = cosine(g, ref_g[r.name]); rel_all.append(np.linalg.norm(g - ref_g[r.name]) / norms[r.name])
        (cos_bn if r.name.startswith("batch_normalization") else cos_conv if r.name.startswith("conv2d") else cos_rest).append(c)
    row["cos_conv_min"], row["cos_bn_min"], row["cos_rest_min"] = min(cos_conv), min(cos_bn), min(cos_rest)
    row["grad_rel_med"], row["grad_rel_max"] = float(np.median(rel_all)), float(max(rel_all))
    flat_ref = np.concatenate([ref_g[r.name].ravel() for r in model.store.refs])
    flat = np.concatenate([r.g.cpu().numpy().ravel() for r in model.store.refs])
    row["grad_L2_rel"] = float(np.linalg.norm(flat - flat_ref) / np.linalg.norm(flat_ref))
    rng_info = None
    if getattr(model.trunk, "half", False):
        mx, sub, nz, tot = 0.0, 0, 0, 0
        for key, t in model.trunk.store.bufs.d.items():
            if t.dtype == torch.float16 and any(s in key for s in ("_gin", "_dc1", "_da1", "_dsc", "res16_")):
                a = t.float().abs()
                mx = max(mx, float(a.max())); nzm = a > 0
                nz += int(nzm.sum()); sub += int((nzm & (a < 6.1e-5)).sum()); tot += a.numel()
        rng_info = dict(scale=model.trunk.gscale, max_abs=mx, headroom=65504.0 / max(mx, 1e-30), subnormal_share=sub / max(nz, 1), zero_share=1 - nz / max(tot, 1))
    return row, rng_info


def main():
    cases = sys.argv[1:] or list(CASES)[:3]
    for case in cases:
        print("== %s" % case)
        keys = None
        for mode in (os.environ.get("NT_MODES", "").split(",") if os.environ.get("NT_MODES") else MODES):
            row, rng_info = run(case, mode)
            if keys is None:
                keys = list(row)
                print("%-9s " % "mode" + " ".join("%13s" % k for k in keys))
            print("%-9s " % mode + " ".join("%13.3e" % row[k] if not k.startswith("cos") else "%13.5f" % row[k] for k in keys))
            if rng_info:
                print("          fp16 gradient range: scale %g, max |g| %.3g (headroom x%.3g to 65504), %.2f %% of the non-zero values subnormal, %.2f %% zeros"
                      % (rng_info["scale"], rng_info["max_abs"], rng_info["headroom"], 100 * rng_info["subnormal_share"], 100 * rng_info["zero_share"]))


if __name__ == "__main__":
    main()
