#!/usr/bin/env python
"""Diagnostic (not collected by pytest): per-tensor deviation of the GPU step from the float64 oracle for both conv math
modes on a small config.  Lives under tests/ because only tests may import oracle/.   python tests/diag_oracle_diff.py"""
import contextlib, io, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model_torch as MT
from aesrc2020_amd import _lib
from aesrc2020_amd.model import SAR_Net

kw = dict(input_shape=[100, 80, 1], res_type="res34", res_filters=16, hidden_dim=64, mto="gvlad", vlad_clusters=4, ghost_clusters=2,
          disc_enable=True, metric_loss="arcface")
B = 4
cfg = MT.Cfg(**kw)
ref = MT.TorchSARNet(cfg, seed=0)
with contextlib.redirect_stdout(io.StringIO()):
    model, _ = SAR_Net(seed=1, use_graph=False, **kw)
model.set_weights_dict({k: v.numpy() for k, v in ref.P.items()})
batch = MT.synthetic_batch(cfg, B, seed=77)
collect = {}
out, total, parts, G = ref.forward_backward(batch)
for mode in ("fp32", "bf16x3"):
    _lib.set_conv_math(mode)
    model.forward_backward_only(batch, batch); torch.cuda.synchronize()
    rows = []
    for r in model.store.refs:
        g = r.g.cpu().double().numpy(); gr = G[r.name].numpy() - (2 * MT.L2 * ref.P[r.name].numpy() if r.name in ref.reg else 0.0)
        m = np.abs(gr).max()
        if m > 1e-6:
            rows.append((float(np.abs(g - gr).max() / m), r.name))
    order = [r.name for r in model.store.refs]
    print("== %s vs float64 oracle: median %.2e; outputs %.2e" % (mode, np.median([e for e, _ in rows]),
          np.abs(model.out["y_accent"].cpu().numpy() - out["y_accent"].detach().numpy()).max()))
    byname = dict((n, e) for e, n in rows)
    # in forward order: where does the error appear going backwards?
    print("   " + "  ".join("%s:%.0e" % (n.replace("batch_normalization", "bn").replace("conv2d", "c").replace("/kernel", "/k").replace("/gamma", "/g").replace("/beta", "/b"), byname[n])
                           for n in order if n in byname and not n.endswith("/bias")))
