#!/usr/bin/env python
"""In-kernel phase timing of the split-precision conv forward kernel (developer tool).

    make -C aesrc2020_amd/csrc timing && python tools/conv_timing.py

Loads tools/libsarnet_timing.so (conv_igemm.hip built with -DSAR_TIMING): thread 0 of every block stamps the shader clock at
phase boundaries.  Prints, per layer shape, the distribution over blocks of: prologue (row decode), first loads, pipeline
fill, k-loop, epilogue (cycles), and the device-wide timeline of block starts/ends (us, 100 MHz real-time counter)."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libsarnet_timing.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.sar_conv2d_fwd_f32.argtypes = [vp] * 7 + [ci] * 12 + [vp]
lib.sar_dbg_set_timing_buffer.argtypes = [vp]

SHAPES = [("s1 3x3 64->64", 50, 10, 64, 64, 3, 1), ("s2 3x3 128->128", 25, 5, 128, 128, 3, 1), ("s3 3x3 256->256", 13, 3, 256, 256, 3, 1)]
B = int(os.environ.get("CB", 64))
for name, H, W, Cin, Cout, k, stride in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(k, k, Cin, Cout, device="cuda") * 0.05
    bias = torch.zeros(Cout, device="cuda"); sc = torch.ones(Cin, device="cuda"); sh = torch.zeros(Cin, device="cuda")
    y = torch.zeros(B, H, W, Cout, device="cuda")
    tb = torch.zeros(1 << 16, 16, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    args = [x.data_ptr(), w.data_ptr(), bias.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, y.data_ptr(), B, H, W, Cin, H, W, Cout, k, k, stride, 1, 1, st]
    fwd, setbuf = lib.sar_conv2d_fwd_f32, lib.sar_dbg_set_timing_buffer
    if os.environ.get("KERNEL", "fwd") == "wgrad":    # weight-gradient kernel instead (phases: row table / first loads / fill / k-loop / atomics)
        dy = torch.randn(B, H, W, Cout, device="cuda"); dw = torch.zeros_like(w)
        lib.sar_conv2d_wgrad_f32.argtypes = [vp] * 6 + [ci] * 12 + [vp]
        args = [x.data_ptr(), dy.data_ptr(), sc.data_ptr(), sh.data_ptr(), dw.data_ptr(), None, B, H, W, Cin, H, W, Cout, k, k, stride, 1, 1, st]
        fwd = lib.sar_conv2d_wgrad_f32
    for it in range(3):
        setbuf(None); torch.cuda.synchronize()
        y.zero_(); fwd(*args)
    torch.cuda.synchronize()
    tb.zero_(); setbuf(tb.data_ptr()); torch.cuda.synchronize()
    y.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fwd(*args); e1.record(); torch.cuda.synchronize()
    t = tb.cpu().numpy().astype(np.int64)
    t = t[t[:, 0] != 0]
    nb = len(t)
    rt0 = t[:, 0].min()
    start_us = (t[:, 0] - rt0) / 100.0; end_us = (t[:, 15] - rt0) / 100.0
    ph = {"prologue": t[:, 2] - t[:, 1], "first loads": t[:, 3] - t[:, 2], "fill(store+sync)": t[:, 4] - t[:, 3],
          "k-loop": t[:, 5] - t[:, 4], "epilogue": t[:, 6] - t[:, 5], "total": t[:, 6] - t[:, 1]}
    print("== %s  B=%d  blocks=%d  event time %.1f us; block starts: p50 %.1f p90 %.1f max %.1f us; ends: p10 %.1f p50 %.1f max %.1f us; block life p50 %.1f us"
          % (name, B, nb, e0.elapsed_time(e1) * 1e3, np.percentile(start_us, 50), np.percentile(start_us, 90), start_us.max(),
             np.percentile(end_us, 10), np.percentile(end_us, 50), end_us.max(), np.percentile(end_us - start_us, 50)))
    for k_, v in ph.items():
        print("   %-18s cycles: p10 %7d  p50 %7d  p90 %7d  max %7d" % (k_, np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
