"""Plan tables of the conv launcher (round 6; csrc/conv_bf16.hip: conv_plan_kernel, include/sarnet.h: sar_conv_plan_workspace_*): a launch that reads the
tables derived from its shape once per shape gives BIT-identical results to one whose blocks derive everything themselves, the workspace may be lent,
detached and lent again, and a workspace that is too small falls back without an error (resnet.py:22-125 -- every conv of the trunk)."""
import ctypes
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
vp, ci = ctypes.c_void_p, ctypes.c_int


class _BN(ctypes.Structure):
    _fields_ = [(n, vp) for n in ("gamma", "beta", "bins", "bbins", "saved", "mm", "mv", "dgamma", "dbeta")] + \
               [(n, ctypes.c_float) for n in ("inv_n", "unbias", "eps", "momentum")] + [("write", ci), ("ginv", ctypes.c_float), ("det", ci), ("pad_", ci)]


# (B, H, W, Cin, Cout, K, stride): every launch class of the trunk -- the 32-channel eight-wave form, the four-wave stride-2 entry and its 1x1 shortcut,
# the BR forms at 64 / 128 / 256 channels, a ragged last tile (B = 3) and the underfilled stage-3 launch (fewer owned pixels than MFMA rows)
SHAPES = [(64, 100, 20, 32, 32, 3, 1), (8, 100, 20, 32, 64, 3, 2), (8, 100, 20, 32, 64, 1, 2), (16, 50, 10, 64, 64, 3, 1), (64, 25, 5, 128, 128, 3, 1),
          (64, 13, 3, 256, 256, 3, 1), (3, 9, 7, 128, 128, 3, 1), (5, 25, 5, 128, 256, 3, 2)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,K,s", SHAPES)
def test_planned_launch_equals_the_unplanned_one_bit_for_bit(B, H, W, Cin, Cout, K, s):
    from aesrc2020_amd import _lib
    from oracle.ops_np import same_pads
    L = _lib.lib().cdll
    L.sar_conv2d_x3.argtypes = [vp] * 8 + [ci] * 13 + [vp, ci, vp, vp, vp, vp, ctypes.c_float, vp]
    L.sar_conv_pack_weights_x3.argtypes = [vp, ci, ci, vp]
    L.sar_conv_plan_workspace_x3.argtypes = [vp, ctypes.c_long]
    L.sar_conv2d_last_class_x3.argtypes = [vp]
    Ho, pt, _ = same_pads(H, K, s); Wo, pl, _ = same_pads(W, K, s)
    torch.manual_seed(Cin + Cout + H)
    x = torch.randn(B, H, W, Cin, device="cuda")
    res = torch.randn(B, Ho, Wo, Cout, device="cuda") if s == 1 and Cin == Cout else None
    w = torch.randn(K, K, Cin, Cout, device="cuda") * 0.05
    gam = torch.rand(Cin, device="cuda") + 0.5; bet = torch.randn(Cin, device="cuda") * 0.1
    n = B * H * W
    mean = x.mean((0, 1, 2)); rstd = 1.0 / torch.sqrt(x.var((0, 1, 2), unbiased=False) + 1e-3)
    saved = torch.stack([mean, rstd, gam * rstd, bet - mean * gam * rstd]).contiguous()
    nf = (Cout // 32) * (2 * K * K * Cin // 16 + 8) * 512; nd = (Cin // 32) * (2 * K * K * Cout // 16 + 8) * 512
    st = torch.cuda.current_stream().cuda_stream
    wf = torch.zeros(nf, dtype=torch.bfloat16, device="cuda"); wd = torch.zeros(nd, dtype=torch.bfloat16, device="cuda")
    tab = torch.tensor([[w.data_ptr(), wf.data_ptr(), wd.data_ptr(), K * K, Cin, Cout, 0, 0]], dtype=torch.int64, device="cuda")
    assert L.sar_conv_pack_weights_x3(tab.data_ptr(), 1, -(-(nf + nd) // 8 // 256), st) == 0
    bn = _BN(gam.data_ptr(), bet.data_ptr(), None, None, saved.data_ptr(), None, None, None, None, 1.0 / n, 1.0, 1e-3, 0.99, 0, 1.0, 1, 0)

    def launch():
        y = torch.full((B, Ho, Wo, Cout), float("nan"), device="cuda"); side = torch.zeros(B, H, W, Cin, device="cuda", dtype=torch.float16)   # (zeros: a strided 1x1 conv's fill leaves the pixels it skips alone)
        bins = torch.zeros(_lib.bn_bins() * 2 * Cout, dtype=torch.int64, device="cuda")
        rc = L.sar_conv2d_x3(x.data_ptr(), None, None, wf.data_ptr(), None, res.data_ptr() if res is not None else None, y.data_ptr(), side.data_ptr(),
                             B, H, W, Cin, Ho, Wo, Cout, K, s, 1, pt, pl, 1, ctypes.byref(bn), 3, bins.data_ptr(), None, None, None, 1.0, st)
        assert rc == 0
        cls = (ci * 6)(); assert L.sar_conv2d_last_class_x3(cls) == 0
        torch.cuda.synchronize()
        return y, side, bins, (cls[5] >> 1) & 1

    own = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
    small = torch.empty(1024, dtype=torch.uint8, device="cuda")
    try:
        assert L.sar_conv_plan_workspace_x3(None, 0) == 0                       # detached: every block derives its tables itself
        y0, s0, b0, p0 = launch()
        assert p0 == 0 and torch.isfinite(y0).all()
        assert L.sar_conv_plan_workspace_x3(own.data_ptr(), own.numel()) == 0   # lent: the first launch writes the shape's tables, the second finds them
        for _ in range(2):
            y1, s1, b1, p1 = launch()
            assert p1 == 1 and torch.equal(y1, y0) and torch.equal(s1, s0) and torch.equal(b1, b0)
        assert L.sar_conv_plan_workspace_x3(small.data_ptr(), small.numel()) == 0   # too small for any shape: the unplanned kernels, no error
        y2, s2, b2, p2 = launch()
        assert p2 == 0 and torch.equal(y2, y0) and torch.equal(b2, b0)
        assert L.sar_conv_plan_workspace_x3(ctypes.c_void_p(own.data_ptr() + 1), 1024) != 0   # unaligned: refused
    finally:
        ws = _lib._PLAN_WS.get("x3")                                             # back to the package's own workspace (or none)
        if ws is not None:
            L.sar_conv_plan_workspace_x3(ws.data_ptr(), ws.numel())
        else:
            L.sar_conv_plan_workspace_x3(None, 0)
