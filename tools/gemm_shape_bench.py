#!/usr/bin/env python
"""Graph-replayed timing of the Dense / GRU-projection GEMM launches at the step's shapes (developer tool): `reps` launches captured in one HIP
graph (no host launch cost between them; operands warm in L2 / MALL, unlike in the step), microseconds per launch."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aesrc2020_amd import _lib as L  # noqa: E402

L.lib(); L.set_gemm_math(os.environ.get("GEMM_MATH", "bf16x3"))
dev = lambda *s: torch.randn(*s, device="cuda")


def timed(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return float(np.median(ts))


def dense(R, N, K, act=1):
    x, W, b, y = dev(R, K), dev(K, N) / K ** 0.5, dev(N), dev(R, N)
    return timed(lambda: L.call("sar_gemm_f32", x, W, y, b, R, N, K, K, N, N, 1, 0, act, 0, 1))


def batched(R, N, K, nb=2):
    x = dev(R, K); Ws = [dev(K, N) for _ in range(nb)]; bs = [dev(N) for _ in range(nb)]; ys = [dev(R, N) for _ in range(nb)]
    gb = L.GemmBatch([(x, Ws[i], ys[i], bs[i]) for i in range(nb)])
    return timed(lambda: L.call("sar_gemm_batched_f32", gb, nb, R, N, K, K, N, N, 1, 0, 0, 0, 1))


def fused(R, K):
    x, W, b, g, be = dev(R, K), dev(K, 256) / K ** 0.5, dev(256), dev(256), dev(256)
    a, y, m, r = dev(R, 256), dev(R, 256), dev(R), dev(R)
    return timed(lambda: L.call("sar_dense_tanh_ln_f32", x, W, b, g, be, a, y, m, r, R, K, 256, 1e-14))


R = int(sys.argv[1]) if len(sys.argv) > 1 else 832
print("rows %d" % R)
for N, K in ((256, 768), (256, 512), (256, 256), (768, 256), (1536, 256)):
    print("  Dense %4d x %4d x %4d: %.2f us (%d tiles)" % (R, N, K, dense(R, N, K), -(-R // 64) * (N // 64)))
print("  GRU projection, both directions in one launch (2 x %d x 768 x 256): %.2f us" % (R, batched(R, 768, 256)))
print("  one direction: %.2f us" % batched(R, 768, 256, 1))
for K in (768, 512, 256):
    print("  Dense + tanh + LN, one launch, K = %d: %.2f us" % (K, fused(R, K)))


# ---- cold launches: what a GEMM costs behind a pass that empties L2 and MALL (as every launch of the step finds them), and how much of that is
# the CODE being cold: {flush, GEMM on operand set 1, the same GEMM on operand set 2}: the second launch finds its instructions, not its data.
def cold(R=832, N=768, K=256, nb=2):
    junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    mk = lambda: (dev(R, K), [dev(K, N) for _ in range(nb)], [dev(N) for _ in range(nb)], [dev(R, N) for _ in range(nb)])
    sets = [mk() for _ in range(3)]
    gbs = [L.GemmBatch([(x, Ws[i], ys[i], bs[i]) for i in range(nb)]) for (x, Ws, bs, ys) in sets]
    call = lambda j: L.call("sar_gemm_batched_f32", gbs[j], nb, R, N, K, K, N, N, 1, 0, 0, 0, 1)
    flush = lambda: L.call("sar_fill_f32", junk.view(torch.float32), 1.0, junk.numel() // 4)
    t_f = timed(flush, 6)
    t_1 = timed(lambda: (flush(), call(0)), 6) - t_f
    t_2 = timed(lambda: (flush(), call(0), call(1)), 6) - t_f
    t_3 = timed(lambda: (flush(), call(0), call(1), call(2)), 6) - t_f
    print("  GRU projection behind a 512 MB fill (%.1f us): first launch %.2f us, + second on other operands %.2f us, + third %.2f us" % (t_f, t_1, t_2 - t_1, t_3 - t_2))


cold()


# ---- cold CODE: four other instantiations of the GEMM kernel (15-22 KB of code each, 64 x 64 x 64 problems: one block) run in front of the timed
# launch and push its instructions out of the 64 KB instruction cache a CU pair shares -- as the ~110 different launches of a step do for each other.
def cold_code(R=832, N=768, K=256, nb=2):
    x, Ws, bs, ys = dev(R, K), [dev(K, N) for _ in range(nb)], [dev(N) for _ in range(nb)], [dev(R, N) for _ in range(nb)]
    gb = L.GemmBatch([(x, Ws[i], ys[i], bs[i]) for i in range(nb)])
    call = lambda: L.call("sar_gemm_batched_f32", gb, nb, R, N, K, K, N, N, 1, 0, 0, 0, 1)
    a, b, c = dev(64 * 256, 64), dev(64, 64), dev(64 * 256, 64)          # 256 row tiles: every CU runs the evicting kernels

    def others():
        L.call("sar_gemm_f32", a, b, c, None, 64 * 256, 64, 64, 64, 64, 64, 1, 1, 0, 0, 1)
        L.call("sar_gemm_f32", a, b, c, None, 64 * 256, 64, 64, 64 * 256, 64, 64, 0, 1, 0, 0, 1)
        L.call("sar_gemm_f32", a, b, c, None, 64 * 256, 64, 64, 64 * 256, 64, 64, 0, 0, 0, 0, 1)
        L.call("sar_gemm_wgrad_f32", a, c, b, None, 64, 64, 64 * 256, 64, 64, 64, 1)
    t_o = timed(others, 6)
    t_c = timed(lambda: (others(), call()), 6) - t_o
    t_w = timed(lambda: (others(), call(), call()), 6) - t_o - t_c
    print("  GRU projection behind four other GEMM instantiations (%.1f us): %.2f us, a second launch right behind it %.2f us" % (t_o, t_c, t_w))


cold_code()
