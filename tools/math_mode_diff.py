#!/usr/bin/env python
"""Gradient deviation of the bf16x3 split-precision conv mode from the exact-fp32 MFMA mode on the full-size workload
(developer tool): per tensor max|g_x3 - g_fp32| / max|g_fp32|, worst tensors first."""
import contextlib, io, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aesrc2020_amd import _lib
from aesrc2020_amd.model import SAR_Net
from aesrc2020_amd.utils import synthetic_batch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
import bench

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]["kw"])
B = int(os.environ.get("CB", 64)); T = int(os.environ.get("CT", 400))
if os.environ.get("CF"):                              # e.g. CF=16 CH=64 CT=100 CB=4: the small parity-test shapes
    wl["res_filters"] = int(os.environ["CF"]); wl["hidden_dim"] = int(os.environ.get("CH", 64))
with contextlib.redirect_stdout(io.StringIO()):
    model, _ = SAR_Net([T, 80, 1], seed=0, use_graph=False, **wl)
xi, yi = synthetic_batch([T, 80, 1], B, wl.get("ctc_enable", False), wl.get("disc_enable", False), seed=5)
res = {}
for mode in ("fp32", "bf16x3", "fp32"):
    _lib.set_conv_math(mode)
    model.forward_backward_only(xi, yi); torch.cuda.synchronize()
    key = mode if mode not in res else mode + "#2"
    res[key] = ({r.name: r.g.detach().cpu().numpy().copy() for r in model.store.refs}, {k: v.detach().cpu().numpy().copy() for k, v in model.out.items()})
def dev(a, b):
    rows = []
    for n in a[0]:
        ga, gb = a[0][n], b[0][n]
        m = np.abs(ga).max()
        rows.append((float(np.abs(ga - gb).max() / m) if m > 0 else 0.0, n, float(m)))
    rows.sort(reverse=True)
    return rows
for other in ("fp32#2", "bf16x3"):
    rows = dev(res["fp32"], res[other])
    print("== fp32 vs %s (fp32#2 = run-to-run noise of split-K atomics): worst tensors" % other)
    for e, n, m in rows[:int(os.environ.get('TOP', 6))]:
        print("   %-40s rel %.2e  (max|g| %.3g)" % (n, e, m))
    print("   median over tensors: %.2e" % np.median([r[0] for r in rows]))
    for k in ("y_accent", "y_disc"):
        if k in res["fp32"][1]:
            print("   output %-10s max abs diff %.2e" % (k, np.abs(res["fp32"][1][k] - res[other][1][k]).max()))
