#!/usr/bin/env python
"""Run-to-run stability of the 2-byte trunk (developer tool): the same forward + backward N times on the same input; every parameter gradient and
the trunk output of every run against the first run's.  The statistic bins are summed by fp32 atomics (differences ~1e-6 relative); anything
past 1e-3 is a race or a read of stale / uninitialised memory.

    python tools/trunk_race.py [mode=bf16x3] [res34] [B=4] [T=200] [N=200]"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from aesrc2020_amd import _lib                                         # noqa: E402
from aesrc2020_amd.engine import ParamStore, store_scope               # noqa: E402
from aesrc2020_amd.resnet import ResNetTrunk                           # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
res = sys.argv[2] if len(sys.argv) > 2 else "res34"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
T = int(sys.argv[4]) if len(sys.argv) > 4 else 200
N = int(sys.argv[5]) if len(sys.argv) > 5 else 200
TOL = float(os.environ.get("RACE_TOL", "1e-3"))                     # SAR_BN_DET=1 RACE_TOL=0: the integer statistic bins make every run bit-identical
_lib.set_trunk_dtype(mode)
store = ParamStore(seed=3)
with store_scope(store):
    tr = ResNetTrunk(res, 32, (T, 80))
store.finalize()
rng = np.random.default_rng(11)
for r in store.refs:
    if r.name.endswith("/gamma"):
        r.w.copy_(torch.tensor(rng.uniform(0.6, 1.4, r.shape), dtype=torch.float32))
    elif r.name.endswith("/beta") or r.name.endswith("/bias"):
        r.w.copy_(torch.tensor(0.2 * rng.standard_normal(r.shape), dtype=torch.float32))
x = torch.tensor(rng.uniform(0, 1, (B, T, 80, 1)), dtype=torch.float32, device="cuda")
junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]     # poison freed memory between runs
ref = None
bad = 0
for it in range(N):
    store.zero_grads()
    out = tr.forward(x, B)
    if it == 0:
        dy = torch.tensor(rng.standard_normal(tuple(out.shape)), dtype=torch.float32, device="cuda")
    tr.backward(dy)
    torch.cuda.synchronize()
    cur = {"out": out.detach().clone()}
    cur.update({r.name: r.g.detach().clone() for r in store.refs})
    if ref is None:
        ref = cur
        continue
    worst = []
    for k, v in cur.items():
        d = float((v - ref[k]).abs().max()); m = float(ref[k].abs().max()) + 1e-20
        if not d <= TOL * m:
            bad += 1
            worst.append((d / m, k))
    if os.environ.get("RACE_VERBOSE"):
        rel = lambda k: float((cur[k] - ref[k]).abs().max()) / (float(ref[k].abs().max()) + 1e-20)
        ks = [k for k in cur if not k.endswith("/bias")]
        ks.sort(key=rel, reverse=True)
        print("run %d: out %.2e; %s" % (it, rel("out"), ", ".join("%s %.2e" % (k.replace("batch_normalization", "bn"), rel(k)) for k in ks[:4])))
    elif worst:
        worst.sort(reverse=True)
        names = [r.name for r in store.refs]
        first = min(names.index(k) for _, k in worst if k in names) if any(k in names for _, k in worst) else -1
        print("run %d: %d tensors off; out %s; worst %s; first (forward order) %s" % (it, len(worst), "OFF" if any(k == "out" for _, k in worst) else "ok",
              ", ".join("%s %.2e" % (k, e) for e, k in worst[:3]), names[first] if first >= 0 else "-"))
print("%d runs, %d tensors off" % (N, bad))
