"""Diagnostic (not collected by pytest; lives under tests/ because only tests may import oracle/): eager vs graph-replayed steps against the oracle (the tool that bisected the hipMemsetAsync-in-graph finding)."""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import model_torch as MT
from aesrc2020_amd.model import SAR_Net
kw = dict(input_shape=[100, 80, 1], res_type="res18", res_filters=16, hidden_dim=64, mto="avg")
cfg = MT.Cfg(**kw)
for use_graph in (False, True, True):
    ref = MT.TorchSARNet(cfg, seed=0)
    with contextlib.redirect_stdout(io.StringIO()):
        model, _ = SAR_Net(seed=1, use_graph=use_graph, **kw)
    model.set_weights_dict({k: v.numpy() for k, v in ref.P.items()})
    for i in range(10):
        b = MT.synthetic_batch(cfg, 2, seed=77 + i)
        p0 = {r.name: r.w.cpu().double().numpy() for r in model.store.refs}
        for k in ref.P: ref.P[k] = torch.tensor(p0[k], dtype=torch.float64)
        out, total, parts, G = ref.forward_backward(b)
        logs = model.train_on_batch(b, b)
        ya = np.abs(model.out["y_accent"].cpu().numpy() - out["y_accent"].detach().numpy()).max()
        print("graph=%s step %d: loss %.5f oracle %.5f  max|dprob| %.2e  l2 %.5f vs %.5f" % (use_graph, i, logs["loss"], float(total), ya, float(model.store.l2_loss.item()), float(parts["l2"])))
