"""Per-stage conv micro-benchmark (B=64, T=400 shapes of res34/F=32): fwd / dgrad / wgrad TFLOP/s via HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aesrc2020_amd import _lib as L
B = int(os.environ.get("CB", "64"))
STAGES = [("s0 3x3 32->32", 100, 20, 32, 32, 3, 1), ("s0 3x3 64->32", 100, 20, 64, 32, 3, 1), ("s1 3x3 64->64", 50, 10, 64, 64, 3, 1),
          ("s1 3x3/2 32->64", 100, 20, 32, 64, 3, 2), ("s2 3x3 128->128", 25, 5, 128, 128, 3, 1), ("s3 3x3 256->256", 13, 3, 256, 256, 3, 1),
          ("s3 1x1/2 128->256", 25, 5, 128, 256, 1, 2)]
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ev0.record()
    for _ in range(iters): fn()
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e-3 / iters
tot = {"fwd": 0, "dgrad": 0, "wgrad": 0}
FILTER = sys.argv[1] if len(sys.argv) > 1 else ""
for name, H, W, ci, co, k, s in STAGES:
    if FILTER and not name.startswith(FILTER):
        continue
    if k == 3:
        Ho, Wo = -(-H // s), -(-W // s); pt = max((Ho - 1) * s + k - H, 0) // 2; pl = max((Wo - 1) * s + k - W, 0) // 2
    else:
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1; pt = pl = 0
    x = torch.randn(B, H, W, ci, device="cuda"); w = torch.randn(k, k, ci, co, device="cuda") * 0.05; b = torch.zeros(co, device="cuda")
    sc = torch.rand(ci, device="cuda") + 0.5; sh = torch.randn(ci, device="cuda") * 0.1
    y = torch.empty(B, Ho, Wo, co, device="cuda"); dy = torch.randn(B, Ho, Wo, co, device="cuda"); dx = torch.empty_like(x); dw = torch.zeros_like(w)
    fl = 2.0 * B * Ho * Wo * co * ci * k * k
    t_f = timed(lambda: L.call("sar_conv2d_fwd_f32", x, w, b, sc, sh, None, y, B, H, W, ci, Ho, Wo, co, k, k, s, pt, pl))
    if ci % 32 == 0 and co % 64 == 0:                # weight planes (sar_conv_weight_split_bf16)
        K_ = k * k * ci
        pl_ = [torch.empty(co, K_, dtype=torch.bfloat16, device="cuda") for _ in range(2)] + [torch.empty(K_, co, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
        nblk = (K_ // 32) * (co // 64)
        table = torch.tensor([[w.data_ptr()] + [t.data_ptr() for t in pl_] + [K_, co, 0]], dtype=torch.int64, device="cuda")
        t_ws = timed(lambda: L.call("sar_conv_weight_split_bf16", table, 1, nblk))
        t_fp = timed(lambda: L.call("sar_conv2d_fwd_ex_f32", x, w, b, sc, sh, None, y, B, H, W, ci, Ho, Wo, co, k, k, s, pt, pl, pl_[0], pl_[1],
                                    None, None, None, None, None, None, None, None, None, 1e-3, 0.99, 1))
        t_dp = timed(lambda: L.call("sar_conv2d_dgrad_ex_f32", dy, w, dx, B, H, W, ci, Ho, Wo, co, k, k, s, pt, pl, 0, pl_[2], pl_[3]))
        print("%-20s weight split %5.1f us | with planes: fwd %6.1f us | dgrad %6.1f us" % (name, t_ws * 1e6, t_fp * 1e6, t_dp * 1e6), flush=True)
    t_d = timed(lambda: L.call("sar_conv2d_dgrad_f32", dy, w, dx, B, H, W, ci, Ho, Wo, co, k, k, s, pt, pl, 0))
    t_w = timed(lambda: L.call("sar_conv2d_wgrad_f32", x, dy, sc, sh, dw, None, B, H, W, ci, Ho, Wo, co, k, k, s, pt, pl))
    print("%-20s GF %.2f | fwd %6.1f us %5.1f TF | dgrad %6.1f us %5.1f TF | wgrad %6.1f us %5.1f TF" %
          (name, fl / 1e9, t_f * 1e6, fl / t_f / 1e12, t_d * 1e6, fl / t_d / 1e12, t_w * 1e6, fl / t_w / 1e12), flush=True)
