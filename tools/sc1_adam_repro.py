#!/usr/bin/env python
"""Reproducer of the write-through anomaly of DESIGN 5.000 (developer tool; VERDICT r5 "Next" item 5).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSAR_ADAM_WT=1 -shared aesrc2020_amd/csrc/optim.hip -o tools/liboptim_wt.so
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSAR_ADAM_WT=2 -shared aesrc2020_amd/csrc/optim.hip -o tools/liboptim_wt2.so
    python tools/sc1_adam_repro.py

Round 5: the Adam sweep with `global_store_dwordx4 ... sc1` (write-through) stores for p, m, v fails tests/test_gpu_ops.py::test_adam_l2_unitnorm at its
second FUSED step, deterministically, and passes again with sc1 loads of the same tensors; tools/wt_probe.hip's 72 patterns never showed a stale read.
This script replays that test's sequence against the write-through build of csrc/optim.hip (-DSAR_ADAM_WT, never part of libsarnet_hip.so) in variants
that take one ingredient away at a time, and says for the first failing step WHICH elements are wrong and WHAT they hold -- the previous step's value
(a lost update / stale read of p) or a value computed from stale moments -- and where their chunks ran (block -> XCD by dispatch order)."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ops_np as O  # noqa: E402  (the float64 Keras-Adam restatement: the checker)

vp, cl = ctypes.c_void_p, ctypes.c_long


def load(path):
    lib = ctypes.CDLL(path)
    lib.sar_adam_prepare.argtypes = [vp, vp]
    lib.sar_adam_l2_step_f32.argtypes = [vp] * 7 + [cl, vp]
    lib.sar_adam_l2_update_f32.argtypes = [vp] * 8 + [cl, vp]
    return lib


def run(lib, n_chunks=40, fused_from=3, steps=6, readback=True, fresh_g=True, tag=""):
    """The sequence of test_adam_l2_unitnorm: `fused_from` two-launch steps (prepare + step), then one-launch steps.  readback: the device tensors
    are copied to the host after every step (the test's check()); fresh_g: the gradient is a NEW device tensor every step (the test's dev(g))."""
    rng = np.random.default_rng(5)
    n = 256 * n_chunks
    p0 = rng.standard_normal(n); g = rng.standard_normal(n) * 0.1
    l2 = np.where(np.arange(n_chunks) % 2 == 0, 1e-4, 0.0); l2e = np.repeat(l2, 256)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    st = torch.zeros(8, dtype=torch.float32, device="cuda")
    st[:5] = torch.tensor([0.01, 2e-4, 0.9, 0.999, 1e-7]); st[7] = 0.5
    p, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    l2l = torch.zeros(1, device="cuda"); tick = torch.zeros(8, dtype=torch.int32, device="cuda")
    gd, l2d = dev(g), dev(l2)
    s = torch.cuda.current_stream().cuda_stream
    pr, mr, vr = p0.copy(), np.zeros(n), np.zeros(n)
    hist = [pr.copy()]
    first_bad = None
    for it in range(steps):
        if fresh_g:
            gd, l2d = dev(g), dev(l2)
        l2l.zero_()
        if it < fused_from:
            lib.sar_adam_prepare(st.data_ptr(), s)
            lib.sar_adam_l2_step_f32(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), l2d.data_ptr(), st.data_ptr(), l2l.data_ptr(), n, s)
        else:
            lib.sar_adam_l2_update_f32(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), l2d.data_ptr(), st.data_ptr(), tick.data_ptr(), l2l.data_ptr(), n, s)
        pr, mr, vr = O.adam_step(pr, 0.5 * g + 2 * l2e * pr, mr, vr, it, lr=0.01)
        hist.append(pr.copy())
        if readback or it == steps - 1:
            got = p.cpu().double().numpy()
            bad = np.abs(got - pr) > 1e-5 * max(1.0, float(np.abs(pr).max()))
            if bad.any() and first_bad is None:
                first_bad = it
                idx = np.nonzero(bad)[0]
                chunks = np.unique(idx // 256)
                stale_prev = np.isclose(got[idx], hist[-2][idx], rtol=0, atol=1e-7).mean()
                # what the kernel would have produced for these elements from the PREVIOUS step's moments and weights, run again
                print("  %s step %d: %d wrong elements in %d of %d chunks (chunks %s ...); %.0f %% of them equal the previous step's p; iterations on device %d; "
                      "max error %.3e" % (tag, it, idx.size, chunks.size, n_chunks, chunks[:8], 100 * stale_prev, int(st.view(torch.int32)[6]), float(np.abs(got - pr)[idx].max())))
                blocks = chunks // 4 % max(1, min((n_chunks + 3) // 4, 2048))
                print("      blocks of those chunks %s -> XCDs (block %% 8) %s" % (np.unique(blocks)[:16], np.unique(blocks % 8)))
    torch.cuda.synchronize()
    return first_bad


if __name__ == "__main__":
    from aesrc2020_amd import _lib
    libs = [("plain stores", load(_lib.LIB_PATH)), ("write-through, bare asm store (round 5's form)", load(os.path.join(HERE, "liboptim_wt.so"))),
            ("write-through, asm store + s_nop 1", load(os.path.join(HERE, "liboptim_wt2.so")))]
    variants = [("the test's sequence", dict()), ("no read-back between steps", dict(readback=False)), ("same gradient tensor every step", dict(fresh_g=False)),
                ("fused from step 0", dict(fused_from=0)), ("never fused", dict(fused_from=99)), ("34 MB (33 280 chunks)", dict(n_chunks=33280, steps=5)),
                ("8 chunks = two blocks", dict(n_chunks=8)), ("4 chunks = one block", dict(n_chunks=4))]
    for name, lib in libs:
        print("== %s" % name)
        for vname, kw in variants:
            fb = run(lib, tag=vname, **kw)
            print("  %-34s first failing step: %s" % (vname, "none" if fb is None else fb))
