"""Timing of sar_accent_classifier_f32 in its three call modes (developer tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aesrc2020_amd import _lib as L
B, H, D1, D2, A = 64, 256, 64, 64, 8
z = lambda *s: torch.randn(*s, device="cuda") * 0.1
x, W1, b1, W2, b2, W3, b3 = z(B, H), z(H, D1), z(D1), z(D1, D2), z(D2), z(D2, A), z(A)
y = torch.eye(A, device="cuda")[torch.randint(0, A, (B,), device="cuda")]; dl = torch.full((B,), 1.0 / B, device="cuda")
h1, h2, pr, ls = z(B, D1), z(B, D2), z(B, A), z(B)
g = [z(H, D1), z(D1), z(D1, D2), z(D2), z(D2, A), z(A)]; dx = z(B, H)
modes = {"predict": (None, None), "eval": (y, None), "train": (y, dl)}
for name, (yy, dd) in modes.items():
    fn = lambda: L.call("sar_accent_classifier_f32", x, W1, b1, W2, b2, W3, b3, yy, dd, h1, h2, pr, ls, *g, dx, B, H, D1, D2, A)
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print("%-8s %7.2f us" % (name, e0.elapsed_time(e1) * 100))
