#!/usr/bin/env python
"""In-kernel phase timing of the recompute-form stem (developer tool):

    make -C aesrc2020_amd/csrc timing && python tools/stem_timing.py [B T D C0]

Loads tools/libsarnet_timing_stem.so (stem_bf16.hip built with -DSAR_TIMING): thread 0 of every block stamps the shader clock at
phase boundaries; prints the distribution over blocks and the device-wide timeline of block starts / ends."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from aesrc2020_amd import _lib as L  # noqa: E402  (BnRef struct + geometry only)
from oracle.ops_np import same_pads  # noqa: E402

X3 = os.environ.get("TIMING_X3", "0") == "1"     # the split-precision build (statistics + pooling passes only): hipcc ... -DSAR_TIMING -DSAR_X3=1 -> libsarnet_timing_stem_x3.so
lib = ctypes.CDLL(os.path.join(HERE, "libsarnet_timing_stem_x3.so" if X3 else "libsarnet_timing_stem.so"))
if X3:
    lib.sar_stem16_stats, lib.sar_stem16_pool_fwd = lib.sar_stem16_stats_x3, lib.sar_stem16_pool_fwd_x3
    lib.sar_stem16_bwd_reduce, lib.sar_stem16_wgrad = lib.sar_stem16_bwd_reduce_x3, lib.sar_stem16_wgrad_x3
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.sar_stem16_stats.argtypes = [vp] * 4 + [ci] * 9 + [vp]
lib.sar_stem16_pool_fwd.argtypes = [vp] * (7 if X3 else 6) + [ci] * 12 + [vp]
lib.sar_stem16_bwd_reduce.argtypes = [vp] * 8 + [ci] * 12 + [vp]
lib.sar_stem16_wgrad.argtypes = [vp] * 9 + [ci] * 12 + [vp]
lib.sar_dbg_set_timing_buffer_stem.argtypes = [vp]
lib.sar_stem16_wgrad_workspace_bytes.restype = ctypes.c_size_t

B, T, D, C0 = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 400, 80, 64)
Ho, pt, _ = same_pads(T, 7, 2); Wo, pl, _ = same_pads(D, 7, 2)
Hp, ppt, _ = same_pads(Ho, 3, 2); Wp, ppl, _ = same_pads(Wo, 3, 2)
x = torch.rand(B, T, D, device="cuda"); w = torch.randn(7, 7, 1, C0, device="cuda") * 0.2
b = torch.zeros(C0, device="cuda"); gamma = torch.ones(C0, device="cuda"); beta = torch.zeros(C0, device="cuda")
bins = torch.zeros(16, 2, C0, device="cuda"); bbins = torch.zeros(16, 2, C0, device="cuda"); saved = torch.zeros(4, C0, device="cuda")
mm = torch.zeros(C0, device="cuda"); mv = torch.ones(C0, device="cuda"); dg = torch.zeros(C0, device="cuda"); db = torch.zeros(C0, device="cuda")
pooled = torch.zeros(B, Hp, Wp, C0, dtype=torch.float32 if X3 else torch.bfloat16, device="cuda"); pooled_h = torch.zeros(B, Hp, Wp, C0, dtype=torch.float16, device="cuda"); idx = torch.zeros(B, Hp, Wp, C0, dtype=torch.uint8, device="cuda")
g = torch.randn(B, Hp, Wp, C0, device="cuda").to(torch.bfloat16)
dw = torch.zeros(7, 7, 1, C0, device="cuda"); work = torch.zeros(lib.sar_stem16_wgrad_workspace_bytes(C0) // 4, device="cuda")
R = B * Ho * Wo
st = torch.cuda.current_stream().cuda_stream


def ref(stats):
    r = L.BnRef()
    r.gamma, r.beta, r.saved = gamma.data_ptr(), beta.data_ptr(), saved.data_ptr()
    r.bins = bins.data_ptr() if stats == "fwd" else None
    r.bbins = bbins.data_ptr() if stats == "bwd" else None
    r.moving_mean, r.moving_var, r.dgamma, r.dbeta = mm.data_ptr(), mv.data_ptr(), dg.data_ptr(), db.data_ptr()
    r.inv_n, r.unbias, r.eps, r.momentum, r.write = 1.0 / R, R / (R - 1.0), 1e-3, 0.99, 1
    return r


conv = (B, T, D, Ho, Wo, C0, pt, pl); pool = (Hp, Wp, ppt, ppl)
P = lambda t: t.data_ptr()
rf, rb = ref("fwd"), ref("bwd")
runs = [
    ("stats", lambda: lib.sar_stem16_stats(P(x), P(w), P(b), P(bins), *conv, 0, st),
     {"stage": (1, 2), "m-tiles": (2, 3), "reduce": (3, 4), "flush": (4, 5), "total": (1, 5)}),
    ("pool_fwd", lambda: (lib.sar_stem16_pool_fwd(P(x), P(w), P(b), ctypes.byref(rf), P(pooled), P(pooled_h), P(idx), *conv, *pool, st) if X3 else
                          lib.sar_stem16_pool_fwd(P(x), P(w), P(b), ctypes.byref(rf), P(pooled), P(idx), *conv, *pool, st)),
     {"stage+prologue": (1, 2), "conv pass 0": (2, 3), "pool pass 0": (3, 4), "conv pass 1": (4, 5), "pool pass 1": (5, 6), "total": (1, 6)}),
    ("wgrad", lambda: (lib.sar_stem16_bwd_reduce(P(g), P(pooled), P(idx), P(x), P(w), P(b), ctypes.byref(rb), P(work), *conv, *pool, st),
                       lib.sar_stem16_wgrad(P(x), P(w), P(b), P(g), P(pooled), P(idx), ctypes.byref(rb), P(dw), P(work), *conv, *pool, st))[1],
     {"prologue": (1, 2), "tile loop": (2, 3), "merge+store": (3, 4), "total": (1, 4)}),
]
if X3:
    runs = runs[:2]
for name, fn, phases in runs:
    tb = torch.zeros(1 << 14, 16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        lib.sar_dbg_set_timing_buffer_stem(None); torch.cuda.synchronize(); assert fn() == 0
    torch.cuda.synchronize()
    lib.sar_dbg_set_timing_buffer_stem(tb.data_ptr()); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); rc = fn(); e1.record(); torch.cuda.synchronize()
    assert rc == 0, rc
    t = tb.cpu().numpy().astype(np.int64)
    t = t[t[:, 0] != 0]
    rt0 = t[:, 0].min()
    s_us = (t[:, 0] - rt0) / 100.0; e_us = (t[:, 15] - rt0) / 100.0
    print("== %s blocks=%d event %.1f us; block starts p50 %.1f max %.1f us; ends p10 %.1f p50 %.1f max %.1f us; block life p50 %.1f max %.1f us"
          % (name, len(t), e0.elapsed_time(e1) * 1e3, np.percentile(s_us, 50), s_us.max(), np.percentile(e_us, 10), np.percentile(e_us, 50), e_us.max(),
             np.percentile(e_us - s_us, 50), (e_us - s_us).max()))
    for k_, (a, c) in phases.items():
        v = t[:, c] - t[:, a]
        print("   %-16s cycles: p10 %7d  p50 %7d  p90 %7d  max %7d" % (k_, np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
    if name == "wgrad":
        n = np.maximum(t[:, 10], 1)
        for nm, sl in (("stage / tile", 8), ("m-tiles / tile", 9)):
            v = t[:, sl] // n
            print("   %-16s cycles: p10 %7d  p50 %7d  p90 %7d  max %7d" % (nm, np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
        print("   tiles per block: min %d p50 %d max %d" % (t[:, 10].min(), np.percentile(t[:, 10], 50), t[:, 10].max()))
