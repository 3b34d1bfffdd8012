"""Kernel time of the accent-branch heads (GPU box): the accent classifier, the margin head's training step, and the one-launch form of
both (csrc/heads_fused.hip), each timed with events over 200 back-to-back calls.  SAR_LIB_PATH selects the build."""
import numpy as np, torch
from aesrc2020_amd import _lib as L

B, H, D1, D2, A = 64, 256, 64, 64, 8
rng = np.random.default_rng(0)
dev = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
zeros = lambda *s: torch.zeros(*s, device="cuda")
x = dev(rng.standard_normal((B, H))); Ws = [dev(rng.standard_normal(s) / np.sqrt(s[0])) for s in ((H, D1), (D1, D2), (D2, A))]
bs = [dev(0.1 * rng.standard_normal(n)) for n in (D1, D2, A)]
y = dev(np.eye(A)[rng.integers(0, A, B)]); dl = dev(rng.uniform(0.5, 1.5, B) / B)
Wm = dev(rng.standard_normal((H, A)) / 16)
d = (x, Ws[0], bs[0], Ws[1], bs[1], Ws[2], bs[2])
o = (zeros(B, D1), zeros(B, D2), zeros(B, A), zeros(B)); g = [zeros(H, D1), zeros(D1), zeros(D1, D2), zeros(D2), zeros(D2, A), zeros(A)]
dx = zeros(B, H); mp, ml, miw, dWm, ws, ws1 = zeros(B, A), zeros(B), zeros(A), zeros(H, A), zeros(H, A), zeros(H * A + 1)


def timed(f, n=200):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            f()
    gr.replay(); torch.cuda.synchronize()
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


acc = lambda: L.call("sar_accent_classifier_f32", *d, y, dl, *o, *g, dx, B, H, D1, D2, A)
mar = lambda: L.call("sar_margin_head_train_f32", x, y, y, dl, Wm, mp, ml, miw, dx, dWm, ws, B, H, A, 2, 30.0, 0.3, 1)
both = lambda: L.call("sar_accent_margin_train_f32", *d, y, dl, *o, *g, dx, y, y, dl, Wm, mp, ml, miw, dWm, ws1, B, H, D1, D2, A, 2, 30.0, 0.3)
print("accent classifier %.1f us, margin head (2 launches) %.1f us, one launch %.1f us" % (timed(acc), timed(mar), timed(both)))
