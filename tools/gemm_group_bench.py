"""Timing of the grouped weight-gradient launch on the cfg2 problem set for several split-K choices (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aesrc2020_amd import _lib as L
L.set_gemm_math(os.environ.get("GG_MATH", "bf16x3"))          # GG_MATH=bf16: the single-pass form
z = lambda *s: torch.randn(*s, device="cuda") * 0.1
R = 2496
base = [(R, 768, 256), (R, 512, 256), (64, 8192, 256)] + [(R, 256, 768)] * 4       # (R, in, out)
def run(sk_of):
    probs = []
    for R_, I, N in base:
        probs.append((z(R_, I), z(R_, N), z(I, N), z(N), I, N, R_, I, N, N, sk_of(R_, I, N)))
    g = L.GemmWgradGroup(probs)
    g.launch(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            g.launch()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    fl = sum(2.0 * a * b * c for a, b, c in base)
    us = e0.elapsed_time(e1) * 100
    return us, fl / us / 1e6, g.blocks
from aesrc2020_amd.engine import pick_splitk
print("default pick_splitk: %.1f us %.0f TF/s %d blocks" % run(lambda R_, I, N: pick_splitk(I, N, R_)))
for sk in (1, 2, 4, 8, 13, 16):
    print("splitk %2d (K >= 512):  %.1f us %.0f TF/s %d blocks" % ((sk,) + run(lambda R_, I, N: sk if R_ >= 512 else 1)))
