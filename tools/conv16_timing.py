#!/usr/bin/env python
"""In-kernel phase timing of the bf16 trunk kernels (developer tool).

    make -C aesrc2020_amd/csrc timing && python tools/conv16_timing.py [fwd|dgrad|wgrad]

Loads tools/libsarnet_timing16.so (conv_bf16.hip built with -DSAR_TIMING): thread 0 of every block stamps the shader clock
at phase boundaries.  Per layer shape: distribution over blocks of the phases (cycles) and the device-wide timeline of block
starts / ends (us, 100 MHz real-time counter)."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
X3 = os.environ.get("TIMING_X3", "0") == "1"                   # the split-precision build (make timing builds tools/libsarnet_timing16_x3.so)
lib = ctypes.CDLL(os.environ.get("TIMING16_LIB") or os.path.join(HERE, "libsarnet_timing16_x3.so" if X3 else "libsarnet_timing16.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
if X3:
    lib.sar_conv2d_x3.argtypes = [vp] * 8 + [ci] * 13 + [vp, ci, vp, vp, vp, vp, ctypes.c_float, vp]
    lib.sar_conv_pack_weights_x3.argtypes = [vp, ci, ci, vp]
    _conv = lambda *a: lib.sar_conv2d_x3(*a[:-1], None, 1.0, a[-1])
    lib.sar_conv2d_bf16 = _conv; lib.sar_conv_pack_weights_bf16 = lib.sar_conv_pack_weights_x3
else:
    lib.sar_conv2d_bf16.argtypes = [vp] * 8 + [ci] * 13 + [vp, ci, vp, vp, vp, vp]
    lib.sar_conv_pack_weights_bf16.argtypes = [vp, ci, ci, vp]
lib.sar_dbg_set_timing_buffer16.argtypes = [vp]
if os.environ.get("TIMING_PLAN", "1") == "1":                   # the launch shapes' plan tables (round 6): TIMING_PLAN=0 = every block derives them
    plan_ws = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
    fnp = lib.sar_conv_plan_workspace_x3 if X3 else lib.sar_conv_plan_workspace_bf16
    fnp.argtypes = [vp, ctypes.c_long]
    assert fnp(plan_ws.data_ptr(), plan_ws.numel()) == 0
BF = torch.bfloat16
NPL = 2 if X3 else 1
planes = lambda t: t.float().contiguous() if X3 else t.to(BF)            # split precision: the trunk's tensors are fp32 in HBM
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B = int(os.environ.get("CB", 64))
SHAPES = [("s0 3x3 32->32", 100, 20, 32, 32, 3), ("s1 3x3 64->64", 50, 10, 64, 64, 3), ("s2 3x3 128->128", 25, 5, 128, 128, 3),
          ("s3 3x3 256->256", 13, 3, 256, 256, 3)]
if os.environ.get("TIMING_SHAPES"):                             # e.g. TIMING_SHAPES=s2,s3
    SHAPES = [s for s in SHAPES if s[0].split()[0] in os.environ["TIMING_SHAPES"].split(",")]
st = torch.cuda.current_stream().cuda_stream
cur = lambda: torch.cuda.current_stream().cuda_stream           # (inside a graph capture the launches go to the capturing stream)
for name, H, W, Cin, Cout, k in SHAPES:
    x = planes(torch.randn(B, H, W, Cin, device="cuda")); w = torch.randn(k, k, Cin, Cout, device="cuda") * 0.05
    y = torch.zeros(B, H, W, Cout, device="cuda", dtype=torch.float32 if X3 else BF); dw = torch.zeros_like(w)
    taps = k * k
    nf = (Cout // 32) * (NPL * taps * Cin // 16 + 8) * 512; nd = (Cin // 32) * (NPL * taps * Cout // 16 + 8) * 512
    wf = torch.zeros(nf, dtype=BF, device="cuda"); wd = torch.zeros(nd, dtype=BF, device="cuda")
    tab = torch.tensor([[w.data_ptr(), wf.data_ptr(), wd.data_ptr(), taps, Cin, Cout, 0, 0]], dtype=torch.int64, device="cuda")
    lib.sar_conv_pack_weights_bf16(tab.data_ptr(), 1, -(-(nf + nd) // 8 // 256), st)
    bins = torch.zeros(32, 2, Cout, device="cuda")
    if mode == "wgrad":
        continue
    elif mode == "fwd1":
        # the second conv of a residual block as the train step issues it: BatchNorm + ReLU of the producer in the fill (constants from
        # the statistic bins), the activated operand written back (side), residual add, statistics of the output
        class BN(ctypes.Structure):
            _fields_ = [(n, vp) for n in ("gamma", "beta", "bins", "bbins", "saved", "mm", "mv", "dgamma", "dbeta")] + \
                       [(n, ctypes.c_float) for n in ("inv_n", "unbias", "eps", "momentum")] + [("write", ci), ("ginv", ctypes.c_float), ("det", ci), ("pad_", ci)]
        n = B * H * W
        gam = torch.ones(Cin, device="cuda"); bet = torch.zeros(Cin, device="cuda"); saved = torch.zeros(4, Cin, device="cuda")
        fbins = torch.zeros(32, 2, Cin, device="cuda"); fbins[0, 1] = float(n)      # mean 0, variance 1
        res = planes(torch.randn(B, H, W, Cout, device="cuda")); side = torch.empty(B, H, W, Cin, device="cuda", dtype=torch.float16 if X3 else BF)
        bn = BN(gam.data_ptr(), bet.data_ptr(), fbins.data_ptr(), None, saved.data_ptr(), None, None, None, None, 1.0 / n, 1.0, 1e-3, 0.99, 0, 1.0, 0, 0)
        keepalive = (gam, bet, saved, fbins, res, side, bn)
        fn = lambda: lib.sar_conv2d_bf16(x.data_ptr(), None, None, wf.data_ptr(), None, res.data_ptr(), y.data_ptr(), side.data_ptr(), B, H, W, Cin, H, W, Cout, k, 1, 1, 1, 1,
                                         1, ctypes.byref(bn), 1, bins.data_ptr(), None, None, cur())
    else:
        wp = wf if mode == "fwd" else wd
        fn = lambda: lib.sar_conv2d_bf16(x.data_ptr(), None, None, wp.data_ptr(), None, None, y.data_ptr(), None, B, H, W, Cin, H, W, Cout, k, 1, 1, 1, 1,
                                         0, None, 1, bins.data_ptr(), None, None, cur())
    if mode != "wgrad":
        names = {"  entry->epi loads": (1, 11), "  ->dma issued": (11, 12), "  ->row table": (12, 13), "  ->bn prologue": (13, 14), "  ->barrier": (14, 2), "  (probe build: 14 -> first barrier passed)": (14, 8), "  (probe build: -> constants done)": (8, 2),
                 "  [14 -> barrier 1 passed]": (14, 16), "  [-> fill set-up]": (16, 17), "  [-> first loads issued]": (17, 18), "  [-> bins summed]": (18, 19), "  [-> barrier 2]": (19, 20),
                 "  [-> finalize]": (20, 21), "  [-> barrier 3]": (21, 2), "  [fill: -> constants read]": (2, 22), "  [fill: -> trip 0 done]": (22, 23), "  [fill: -> rest]": (23, 3),
                 "prologue": (1, 2), "fill": (2, 3), "  k first half": (4, 8), "fill sync": (3, 4), "k-phase": (4, 5), "  epi: loads + barrier": (5, 10), "  epi: staging + barrier": (10, 9), "  epi: row pass": (9, 6), "epilogue": (5, 6), "stats": (6, 7), "total": (1, 7)}
    tb = torch.zeros(1 << 16, 32, dtype=torch.int64, device="cuda")
    for it in range(3):
        lib.sar_dbg_set_timing_buffer16(None); torch.cuda.synchronize(); fn()
    torch.cuda.synchronize()
    if os.environ.get("TIMING_NOSTAMP"):                          # launch time without the stamps' own cost: 20 launches in a captured graph, as the step runs them
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        print("== %s %s B=%d: %.2f us per launch (20 launches in one graph, no stamps)" % (mode, name, B, e0.elapsed_time(e1) * 1e3 / 20))
        continue
    lib.sar_dbg_set_timing_buffer16(tb.data_ptr()); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); rc = fn(); e1.record(); torch.cuda.synchronize()
    assert rc == 0, rc
    t = tb.cpu().numpy().astype(np.int64)
    t = t[t[:, 0] != 0]
    rt0 = t[:, 0].min()
    start_us = (t[:, 0] - rt0) / 100.0; end_us = (t[:, 15] - rt0) / 100.0
    print("== %s %s B=%d blocks=%d event %.1f us; block starts p50 %.1f max %.1f us; ends p10 %.1f p50 %.1f max %.1f us; block life p50 %.1f us"
          % (mode, name, B, len(t), e0.elapsed_time(e1) * 1e3, np.percentile(start_us, 50), start_us.max(), np.percentile(end_us, 10),
             np.percentile(end_us, 50), end_us.max(), np.percentile(end_us - start_us, 50)))
    if mode == "wgrad":
        for nm, sl in (("loop: wait", 8), ("loop: barrier", 9), ("loop: issue", 10), ("loop: compute", 11)):
            v = t[:, sl]
            print("   %-12s cycles: p10 %7d  p50 %7d  p90 %7d  max %7d (sum over the block's chunks, wave 0)" % (nm, np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
    for k_, (a, b) in names.items():
        v = t[:, b] - t[:, a]
        print("   %-12s cycles: p10 %7d  p50 %7d  p90 %7d  max %7d" % (k_, np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))

if mode == "wgrad":
    # the batched 3x3 weight gradient as the train step launches it: all stride-1 convs of stages 1-3 (64 x 64 tiles) in one plan
    class WC(ctypes.Structure):
        _fields_ = [(n, vp) for n in ("x", "dy", "dw", "dbias")] + [(n, ci) for n in ("B", "H", "W", "Cin", "Ho", "Wo", "Cout", "K", "s", "pt", "pl", "pad_")]
    lib.sar_wgrad_plan_bf16.argtypes = [vp, ci, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp]
    lib.sar_conv2d_wgrad_batch_bf16.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, vp]
    for label, tile, target, layers in (("stages 1-3, 64x64 tiles", (2, 2), 640, [(50, 10, 64, 7), (25, 5, 128, 11), (13, 3, 256, 5)]),
                                        ("stages 1-2, 64x64 tiles", (2, 2), 448, [(50, 10, 64, 7), (25, 5, 128, 11)]),
                                        ("stage 3, 64x64 tiles", (2, 2), 192, [(13, 3, 256, 5)]),
                                        ("stage 0, 32x32 tiles", (1, 1), 768, [(100, 20, 32, 6)])):
        keep, descs = [], []
        for H, W, C, cnt in layers:
            for _ in range(cnt):
                x = torch.randn(B, H, W, C, device="cuda").to(BF); dy = torch.randn(B, H, W, C, device="cuda").to(BF)
                dw = torch.zeros(3, 3, C, C, device="cuda"); keep += [x, dy, dw]
                descs.append((x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, B, H, W, C, H, W, C, 3, 1, 1, 1, 0))
        arr = (WC * len(descs))(*[WC(*d) for d in descs])
        cbuf = ctypes.create_string_buffer(len(descs) * 128); ibuf = ctypes.create_string_buffer(16384 * 32)
        ni, l4 = ctypes.c_int(0), (ctypes.c_int * 8)()
        tbuf = ctypes.create_string_buffer(16384 * 32); sbuf = (ctypes.c_int * 16384)()
        assert lib.sar_wgrad_plan_bf16(arr, len(descs), target, tile[0], tile[1], cbuf, ibuf, 16384, ctypes.byref(ni), l4, tbuf, sbuf) == 0
        td = torch.frombuffer(bytearray(tbuf.raw[:max(l4[4], 1) * 32]), dtype=torch.uint8).cuda(); sd = torch.tensor(list(sbuf)[:max(l4[5], 1)], dtype=torch.int32).cuda()
        slab = torch.empty(max(l4[5] * l4[6], 1), device="cuda")
        cd = torch.frombuffer(bytearray(cbuf.raw), dtype=torch.uint8).cuda(); idv = torch.frombuffer(bytearray(ibuf.raw[:ni.value * 32]), dtype=torch.uint8).cuda()
        fn = lambda: lib.sar_conv2d_wgrad_batch_bf16(cd.data_ptr(), idv.data_ptr(), ni.value, 3, l4, td.data_ptr(), sd.data_ptr(), slab.data_ptr(), st)
        tb = torch.zeros(1 << 14, 32, dtype=torch.int64, device="cuda")
        for it in range(3):
            lib.sar_dbg_set_timing_buffer16(None); torch.cuda.synchronize(); fn()
        torch.cuda.synchronize()
        lib.sar_dbg_set_timing_buffer16(tb.data_ptr()); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = fn(); e1.record(); torch.cuda.synchronize()
        assert rc == 0, rc
        t = tb.cpu().numpy().astype(np.int64)[:ni.value]
        rt0 = t[:, 0].min()
        s_us = (t[:, 0] - rt0) / 100.0; e_us = (t[:, 15] - rt0) / 100.0
        print("== wgrad batched, %s: items=%d launch4=%s event %.1f us; starts p50 %.1f p90 %.1f max %.1f; ends p50 %.1f max %.1f us"
              % (label, ni.value, list(l4), e0.elapsed_time(e1) * 1e3, np.percentile(s_us, 50), np.percentile(s_us, 90), s_us.max(), np.percentile(e_us, 50), e_us.max()))
        nch = np.maximum(t[:, 10], 1)
        for nm, v in (("setup (first DMA issued)", t[:, 2] - t[:, 1]), ("loop", t[:, 3] - t[:, 2]), ("flush", t[:, 4] - t[:, 3]),
                      ("loop / chunk", (t[:, 3] - t[:, 2]) / nch), ("wait+barrier / chunk", t[:, 8] / nch), ("issue / chunk", t[:, 9] / nch), ("chunks per item", t[:, 10])):
            print("   %-26s cycles: p10 %8d  p50 %8d  p90 %8d  max %8d" % (nm, np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
