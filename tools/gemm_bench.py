"""Isolated timing of the model's Dense / GRU-projection GEMMs (developer tool): graph replays of 20 launches per shape, so that what a
launch costs inside the train step (tools/step_timeline.py) can be set against what it costs with warm caches.
    python tools/gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aesrc2020_amd import _lib as L
from aesrc2020_amd._lib import call
L.set_gemm_math(os.environ.get("GEMM_MATH", "bf16"))
z = lambda *s: torch.randn(*s, device="cuda") * 0.1
R = 2496
# (label, M, N, K, a_kc, b_kc, act)   forward: C = tanh(A (M,K) . B (K,N) + b); input gradient: dX = dY (M,K) . W^T, W stored (N,K)
CASES = [("fwd 256->256 tanh", R, 256, 256, 1, 0, 1), ("fwd 256->768", R, 768, 256, 1, 0, 0), ("fwd 512->256 tanh", R, 256, 512, 1, 0, 1),
         ("dgrad 256<-256", R, 256, 256, 1, 1, 0), ("dgrad 256<-768", R, 256, 768, 1, 1, 0), ("dgrad 512<-256", R, 512, 256, 1, 1, 0)]
if os.environ.get("GEMM_KSWEEP"):                                   # fixed cost vs cost per 64-wide k-tile
    CASES = [("dgrad K=%d" % k, R, 256, k, 1, 1, 0) for k in (64, 128, 256, 512, 1024)] + [("fwd K=%d" % k, R, 256, k, 1, 0, 0) for k in (64, 256, 1024)]
for label, M, N, K, akc, bkc, act in CASES:
    A = z(M, K); B = z(N, K) if bkc else z(K, N); C = torch.empty(M, N, device="cuda"); bias = z(N)
    fn = lambda: call("sar_gemm_f32", A, B, C, bias, M, N, K, K, K if bkc else N, N, akc, bkc, act, 0, 1)
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("%-20s M=%d N=%d K=%d  %.2f us per launch  %.1f TFLOP/s" % (label, M, N, K, us, 2.0 * M * N * K / us / 1e6))
