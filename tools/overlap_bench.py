"""Do dgrad and wgrad of one conv layer overlap on two HIP streams?  (sequential vs concurrent wall time)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aesrc2020_amd import _lib as L
B = 64
for name, H, W, ci, co in [("s0", 100, 20, 32, 32), ("s1", 50, 10, 64, 64), ("s2", 25, 5, 128, 128), ("s3", 13, 3, 256, 256)]:
    k, s, pt, pl = 3, 1, 1, 1
    x = torch.randn(B, H, W, ci, device="cuda"); w = torch.randn(k, k, ci, co, device="cuda") * 0.05
    dy = torch.randn(B, H, W, co, device="cuda"); dx = torch.empty_like(x); dw = torch.zeros_like(w)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def dgrad(st): L.call("sar_conv2d_dgrad_f32", dy, w, dx, B, H, W, ci, H, W, co, k, k, s, pt, pl, 0, stream=st.cuda_stream)
    def wgrad(st): L.call("sar_conv2d_wgrad_f32", x, dy, None, None, dw, None, B, H, W, ci, H, W, co, k, k, s, pt, pl, stream=st.cuda_stream)
    def run(conc, iters=30):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        for _ in range(iters):
            if conc:
                ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
                dgrad(s1); wgrad(s2)
                ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
            else:
                dgrad(s1); wgrad(s1)
        e1.record(s1); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters
    run(False, 3); run(True, 3)
    print("%s: sequential %.1f us, two streams %.1f us" % (name, run(False), run(True)), flush=True)
