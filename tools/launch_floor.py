#!/usr/bin/env python
"""What one dependent kernel boundary costs on this stack.

    make -C aesrc2020_amd/csrc tools && python tools/launch_floor.py

MI355X_MICROARCH.md's price list ("boundary") measures 1.45 us between trivial 256-workgroup kernels and 1.7-1.9 us between
streaming kernels, eager == hipGraph.  DESIGN.md (round 2) quoted ~4.5 us per launch.  This tool replays captured graphs of N
launches of: an empty kernel in several launch shapes (threads, grid, kernel-argument bytes, dynamic LDS), a one-round-trip
kernel, a streaming copy, fixed-duration spin bodies, the library's own conv_bf16_kernel<2, 8, 9, 2> (stage-2 3x3, forward), and a
stamp kernel whose blocks record the device clock at start and end (where a boundary's time goes: ramp, drain, gap)."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
lf = ctypes.CDLL(os.path.join(HERE, "liblaunch_floor.so"))
vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
lf.lf_empty.argtypes = [ci, ci, vp]; lf.lf_empty_big.argtypes = [ci, ci, vp]; lf.lf_lds.argtypes = [ci, ci, ci, vp]
lf.lf_touch.argtypes = [vp, ci, ci, vp]; lf.lf_stream.argtypes = [vp, vp, cl, ci, vp]; lf.lf_spin.argtypes = [ci, ci, ci, vp]
lf.lf_stamp.argtypes = [vp, ci, ci, ci, ci, ci, vp]
lf.lf_eager_chain.argtypes = [ci, ci, ci, vp]; lf.lf_eager_chain.restype = ctypes.c_float
lf.lf_graph_chain.argtypes = [ci, ci, ci, ci, vp]; lf.lf_graph_chain.restype = ctypes.c_float


def stream():
    return torch.cuda.current_stream().cuda_stream


def graph_us(fn, n, reps=5):
    """us per launch of a captured graph of n launches (fn(i) issues launch i on the current stream)."""
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    torch.cuda.set_device(0)
    N = 200
    rows = []
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        s = stream()
        rows.append(("eager C loop, empty 256x256", lf.lf_eager_chain(2000, 256, 256, s)))
        rows.append(("explicit hipGraph (no capture), empty 256x256", lf.lf_graph_chain(N, 256, 256, 5, s)))
        rows.append(("captured graph, empty 1x64", graph_us(lambda i: lf.lf_empty(1, 64, stream()), N)))
        rows.append(("captured graph, empty 256x256", graph_us(lambda i: lf.lf_empty(256, 256, stream()), N)))
        rows.append(("captured graph, empty 256x512", graph_us(lambda i: lf.lf_empty(256, 512, stream()), N)))
        rows.append(("captured graph, empty 1024x256", graph_us(lambda i: lf.lf_empty(1024, 256, stream()), N)))
        rows.append(("captured graph, empty 256x256 + 320 B of kernel arguments", graph_us(lambda i: lf.lf_empty_big(256, 256, stream()), N)))
        for lds in (64 * 1024, 120 * 1024, 160 * 1024):
            rows.append(("captured graph, 256x512 + %d KB dynamic LDS" % (lds // 1024), graph_us(lambda i: lf.lf_lds(256, 512, lds, stream()), N)))
        p = torch.zeros(256 * 256, device="cuda")
        rows.append(("captured graph, one load+store round trip 256x256", graph_us(lambda i: lf.lf_touch(p.data_ptr(), 256, 256, stream()), N)))
        a = torch.empty(32 << 20, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
        t_copy = graph_us(lambda i: lf.lf_stream((a if i % 2 == 0 else b).data_ptr(), (b if i % 2 == 0 else a).data_ptr(), (32 << 20) // 16, 2048, stream()), 50)
        rows.append(("captured graph, 32 MB streaming copy (2048x256), per launch", t_copy))
        for cyc, nm in ((12000, "~5 us"), (24000, "~10 us")):
            t1 = graph_us(lambda i: lf.lf_spin(256, 256, cyc, stream()), N)
            rows.append(("captured graph, spin body %s (%d cycles) 256x256: per launch" % (nm, cyc), t1))
        # stamp chain: 256 blocks x 512 threads x 150 KB LDS (one per CU, the conv's shape) and 256 x 256 x 0 LDS
        for thr, lds, nm in ((256, 0, "256 thr, no LDS"), (512, 150 * 1024, "512 thr, 150 KB LDS (one block per CU)")):
            for grid in (256, 250):
                nl = 40
                buf = torch.zeros(nl * grid * 2, dtype=torch.int64, device="cuda")
                t = graph_us(lambda i: lf.lf_stamp(buf.data_ptr(), i % nl, 12000, grid, thr, lds, stream()), nl, reps=3)
                v = buf.cpu().numpy().reshape(nl, grid, 2).astype(np.float64) / 100.0       # us
                ramp = np.median(v[:, :, 0].max(1) - v[:, :, 0].min(1)); drain = np.median(v[:, :, 1].max(1) - v[:, :, 1].min(1))
                gap = np.median(v[1:, :, 0].min(1) - v[:-1, :, 1].max(1)); life = np.median(v[:, :, 1] - v[:, :, 0])
                rows.append(("stamp chain %s, grid %d: per launch" % (nm, grid), t))
                rows.append(("   block life %.2f, start ramp %.2f, end spread %.2f, last end -> next first start %.2f" % (life, ramp, drain, gap), float("nan")))
        # the library's own dominant conv launch (stage 2, 128 -> 128 3x3 at B = 64, forward with statistics), chained y -> x -> y
        from aesrc2020_amd import _lib
        L = _lib.lib()
        BF = torch.bfloat16
        B, H, W, C = 64, 25, 5, 128
        x = torch.randn(B, H, W, C, device="cuda").to(BF); y = torch.empty_like(x)
        w = torch.randn(3, 3, C, C, device="cuda") * 0.03
        nf = (C // 32) * (9 * C // 16 + 8) * 512
        wf = torch.zeros(nf, dtype=BF, device="cuda"); wd = torch.zeros(nf, dtype=BF, device="cuda")
        tab = torch.tensor([[w.data_ptr(), wf.data_ptr(), wd.data_ptr(), 9, C, C, 0, 0]], dtype=torch.int64, device="cuda")
        _lib.call("sar_conv_pack_weights_bf16", tab, 1, -(-(2 * nf) // 8 // 256))
        bins = torch.zeros(16 * 2 * C, device="cuda")

        def conv(i, epi=1):
            src, dst = (x, y) if i % 2 == 0 else (y, x)
            _lib.call("sar_conv2d_bf16", src, None, None, wf, None, None, dst, None, B, H, W, C, H, W, C, 3, 1, 1, 1, 1, 0, None, epi,
                      bins if epi else None, None, None)
        t_conv = graph_us(conv, N)
        rows.append(("captured graph, conv_bf16_kernel<2,8,9,2> stage-2 forward (fill 0, statistics): per launch", t_conv))
        rows.append(("captured graph, the same without the statistics epilogue: per launch", graph_us(lambda i: conv(i, 0), N)))

        def conv_empty(i):
            conv(i); lf.lf_empty(256, 256, stream())
        rows.append(("captured graph, conv + empty 256x256 pairs: per PAIR", graph_us(conv_empty, N // 2)))
    w_ = max(len(r[0]) for r in rows)
    for name, v in rows:
        print(("%-" + str(w_) + "s  %s") % (name, "" if v != v else "%7.2f us" % v))


if __name__ == "__main__":
    main()
