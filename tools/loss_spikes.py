#!/usr/bin/env python
"""How often does a training run of the convergence task spike on its loss floor, by arithmetic?  (developer tool; round 6)

    python tools/loss_spikes.py [runs]

tests/test_gpu_convergence.py's task at hidden_dim 256, 400 Adam steps, from the same initial weights: the fp32-class reference (round 1's kernels,
"bf16x3"), the package default ("x3": split-precision forward, single-pass fp16 backward under the gradient scale, fp16 MFMA recurrence) and the default
with the three-pass backward restored (SAR_X3_BWD=x3 SAR_GEMM_BWD=x3).  Prints, per run, the steps of 200-400 beyond twice the median of those steps,
the largest of them, and the median itself; then the totals per arithmetic."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("conv", os.path.join(os.path.dirname(HERE), "tests", "test_gpu_convergence.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pool = m.make_pool()
tot = {}
for rep in range(runs):
    ref_loss, _, w0 = m.train("bf16x3", pool, hidden=256)
    curves = [("reference (fp32-class, round-1 kernels)", ref_loss)]
    curves.append(("default (x3 forward, fp16 backward)", m.train("x3", pool, weights=w0, hidden=256)[0]))
    os.environ["SAR_X3_BWD"] = "x3"; os.environ["SAR_GEMM_BWD"] = "x3"
    try:
        curves.append(("x3 forward, three-pass backward", m.train("x3", pool, weights=w0, hidden=256)[0]))
    finally:
        del os.environ["SAR_X3_BWD"]; del os.environ["SAR_GEMM_BWD"]
    if os.environ.get("LOSS_SPIKES_EXTRA"):
        # which half of the forward: the split-precision trunk with fp32 GEMMs / recurrence, and round 1's trunk with split-precision GEMMs + fp16 MFMA recurrence
        m.MODES["x3_trunk_only"] = ("bf16x3", "bf16x3", "fp32"); m.MODES["x3_gemm_only"] = ("fp32", "bf16x3", "bf16x3")
        curves.append(("x3 trunk, fp32 GEMMs and recurrence", m.train("x3_trunk_only", pool, weights=w0, hidden=256)[0]))
        curves.append(("round-1 trunk, x3 GEMMs + fp16 recurrence", m.train("x3_gemm_only", pool, weights=w0, hidden=256)[0]))
    for name, v in curves:
        tail = v[200:]; med = float(np.median(tail)); n = int((tail > 2 * med).sum())
        t = tot.setdefault(name, [0, 0, 0.0]); t[0] += n; t[1] += int(n >= 10); t[2] = max(t[2], float(tail.max()))
        print("run %d  %-42s median %.3f  steps beyond 2 x median: %3d  max %.1f" % (rep, name, med, n, tail.max()))
for name, (n, ep, mx) in tot.items():
    print("%-42s %d spiking steps in %d runs, %d runs with ten or more, largest loss %.1f" % (name, n, runs, ep, mx))
