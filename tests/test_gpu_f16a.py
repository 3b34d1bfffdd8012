"""GPU parity tests of the MIXED BACKWARD's conv launches (round 5; csrc/conv_bf16.hip template parameter A32, entry points sar_conv2d_f16a /
sar_conv2d_pair_f16a): the fp16 build's input-gradient kernels reading the two tensors of the FORWARD pass a gradient launch touches -- the
BatchNorm input of the fill's backward transform (src2) and the BatchNorm input of the epilogue's backward sums (e_y) -- as the fp32 tensors the
split-precision forward (sar_conv2d_x3) left in HBM, everything that is a gradient as fp16.  Reference: oracle/ops_np.py (float64) with gradients
and kernels drawn on / rounded to the fp16 grid and the forward tensors as arbitrary fp32 values; a stored fp16 tensor within ONE ulp (1e-3 of
the reference's max-abs: with forward operands off the fp16 grid the kernel's fp32 transform and the float64 one land on different sides of a
rounding boundary now and then; tests/test_gpu_bf16.py, every operand on the grid, holds the fp16 build to half an ulp), fp32 sums within 2e-4.  Replaces, in the package's default mode, the reference's
Conv2D input gradients (resnet.py:39-42, 60-63, 82-87, 113-117) behind the BatchNormalization backward (resnet.py:22-26)."""
import numpy as np
import pytest
import torch

from oracle import ops_np as O
from tests import test_gpu_bf16 as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def fp16_build():
    old = (T.BF, T.SFX, T.TOL_BF, T.X3)
    T.BF, T.SFX, T.TOL_BF, T.X3 = torch.float16, "f16", 1e-3, False
    yield
    T.BF, T.SFX, T.TOL_BF, T.X3 = old


@pytest.fixture(scope="module")
def L():
    from aesrc2020_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.lib()
    return _lib


def f32(a):
    return np.asarray(a, np.float32).astype(np.float64)


def dgrad_a(cv, dy, H, W, dy2=None, fill=0, fbn=None, res=None, e_bins=None, e_y=None, e_saved=None, side=None, dy3=None):
    B, ho, wo, _ = dy.shape[-4:]
    _, _, pt, pl = cv.geo(H, W)
    dx = T.empty16(B, H, W, cv.cin)
    cv.L.call("sar_conv2d_f16a", dy, dy2, dy3, cv.wd, None, res, dx, side, B, ho, wo, cv.cout, H, W, cv.cin, cv.k, 1, cv.s,
              cv.k - 1 - pt, cv.k - 1 - pl, fill, fbn, 2 if e_bins is not None else 0, e_bins, e_y, e_saved)
    return dx


@pytest.mark.parametrize("Cin,Cout,k", [(32, 32, 3), (64, 128, 3), (256, 256, 3), (64, 32, 1), (512, 512, 3)])
def test_split_precision_pack_writes_the_fp16_input_gradient_stream(L, Cin, Cout, k):
    """sar_conv_pack_weights_x3 with bit 0 of the row's last word: the forward stream as hi / lo bf16 pairs (unchanged), the input-gradient stream
    as the single plane of fp16 fragments sar_conv_pack_weights_f16 writes -- bit for bit."""
    rng = np.random.default_rng(Cin + Cout + k)
    w = T.dev(rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin))
    taps = k * k

    def pack(sfx, npf, npd, flag):
        nf = (Cout // 32) * (npf * taps * Cin // 16 + 8) * 512; nd = (Cin // 32) * (npd * taps * Cout // 16 + 8) * 512
        wf = torch.zeros(nf, dtype=torch.int16, device="cuda"); wd = torch.zeros(nd, dtype=torch.int16, device="cuda")
        tab = torch.tensor([[w.data_ptr(), wf.data_ptr(), wd.data_ptr(), taps, Cin, Cout, 0, flag]], dtype=torch.int64, device="cuda")
        L.call("sar_conv_pack_weights_" + sfx, tab, 1, -(-(nf + nd) // 8 // 256))
        torch.cuda.synchronize()
        return wf.cpu(), wd.cpu()

    wf_x3, wd_x3 = pack("x3", 2, 2, 0)
    wf_mix, wd_mix = pack("x3", 2, 1, 1)
    wf_h, wd_h = pack("f16", 1, 1, 0)
    assert torch.equal(wf_mix, wf_x3) and torch.equal(wd_mix, wd_h)
    assert wd_x3.numel() != wd_mix.numel() and int((wd_mix != 0).sum()) > 0


@pytest.mark.parametrize("shape", T.SOME, ids=T.IDS_SOME)
def test_input_gradient_with_fp32_forward_tensor_in_the_epilogue_sums(L, shape):
    B, H, W, Cin, Cout, k, s, pad = shape
    rng = np.random.default_rng(12 + sum(shape[:7]))
    w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cout)
    cv = T.Conv(L, w, s, pad)
    ho, wo, _, _ = cv.geo(H, W)
    _, cache = O.conv2d_fwd(np.zeros((B, H, W, Cin)), cv.wq, None, (s, s), pad)
    dy = T.grid(rng, (B, ho, wo, Cout))
    ref = O.conv2d_bwd(dy, cache)[0]
    res = T.grid(rng, ref.shape)
    dx = dgrad_a(cv, T.dbf(dy), H, W, res=T.dbf(res))
    T.check(dx, ref + res, T.TOL_BF, "dx + res (no forward tensor involved: the plain form)")
    assert torch.equal(dx, cv.dgrad(T.dbf(dy), H, W, res=T.dbf(res)))               # ... bit for bit what sar_conv2d_f16 gives
    # epilogue sums of the BatchNorm in front of the conv: its input xin is an fp32 tensor of the forward pass (NOT on the fp16 grid)
    bn = T.BN(L, Cin, rng)
    xin = f32(rng.standard_normal((B, H, W, Cin)) * 1.003 + 0.3)
    bn.set_fwd_stats(xin); bn.set_saved()
    dx2 = dgrad_a(cv, T.dbf(dy), H, W, e_bins=bn.d["bbins"], e_y=T.dev(xin), e_saved=bn.d["saved"])
    T.check(dx2, ref, T.TOL_BF, "dx")
    S1, S2, _, _ = bn.bwd_sums(ref, xin)
    tot = T.val(bn.d["bbins"]).sum(0)
    T.check(tot[0], S1, 2e-4, "sum dz"); T.check(tot[1], S2, 2e-4, "sum dz*xhat")


@pytest.mark.parametrize("shape", T.SOME, ids=T.IDS_SOME)
def test_input_gradient_with_fp32_forward_tensor_in_the_fill(L, shape):
    B, H, W, Cin, Cout, k, s, pad = shape
    rng = np.random.default_rng(13 + sum(shape[:7]))
    w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cout)
    cv = T.Conv(L, w, s, pad)
    ho, wo, _, _ = cv.geo(H, W)
    _, cache = O.conv2d_fwd(np.zeros((B, H, W, Cin)), cv.wq, None, (s, s), pad)
    da = T.grid(rng, (B, ho, wo, Cout))
    yb = f32(rng.standard_normal((B, ho, wo, Cout)) * 0.997 + 0.2)                  # the BatchNorm's input: an fp32 tensor of the forward pass
    bn = T.BN(L, Cout, rng)
    bn.set_fwd_stats(yb); bn.set_saved(); bn.set_bwd_stats(da, yb)
    dyq = T.rbf(bn.dx(da, yb))                                                      # what the fill writes to LDS (fp16)
    ref = O.conv2d_bwd(dyq, cache)[0]
    side = T.side16(da.shape)
    dx = dgrad_a(cv, T.dbf(da), H, W, dy2=T.dev(yb), fill=2, fbn=bn.ref(bn.n, "bwd", True), side=side)
    T.check(dx, ref, T.TOL_BF, "dx"); T.check(side, dyq, T.TOL_BF, "side = BatchNorm backward of (da fp16, y fp32)")
    T.check(bn.d["dg"], bn.S2, 1e-5, "dgamma"); T.check(bn.d["db"], bn.S1, 1e-5, "dbeta")
    add = T.grid(rng, da.shape)
    dyq3 = T.rbf(bn.dx(da, yb) + add)
    ref3 = O.conv2d_bwd(dyq3, cache)[0]
    side3 = T.side16(da.shape)
    dx3 = dgrad_a(cv, T.dbf(da), H, W, dy2=T.dev(yb), fill=3, fbn=bn.ref(bn.n, "bwd", False), side=side3, dy3=T.dbf(add))
    T.check(dx3, ref3, T.TOL_BF, "dx (fill 3)"); T.check(side3, dyq3, T.TOL_BF, "side = BatchNorm backward + shortcut gradient")
    # fill 2 WITH the epilogue sums of the BatchNorm in front (what conv1's input-gradient launch of a residual block is): both forward tensors fp32
    bn0 = T.BN(L, Cin, rng)
    xin = f32(rng.standard_normal((B, H, W, Cin)) + 0.1)
    bn0.set_fwd_stats(xin); bn0.set_saved()
    dx4 = dgrad_a(cv, T.dbf(da), H, W, dy2=T.dev(yb), fill=2, fbn=bn.ref(bn.n, "bwd", False), e_bins=bn0.d["bbins"], e_y=T.dev(xin), e_saved=bn0.d["saved"])
    assert torch.equal(dx4, dx)
    S1, S2, _, _ = bn0.bwd_sums(ref, xin)
    tot = T.val(bn0.d["bbins"]).sum(0)
    T.check(tot[0], S1, 3e-4, "sum dz"); T.check(tot[1], S2, 3e-4, "sum dz*xhat")


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(64, 50, 10, 64, 128), (64, 25, 5, 128, 256), (5, 9, 7, 64, 128)])
def test_paired_input_gradients(L, B, H, W, Cin, Cout):
    """sar_conv2d_pair_f16a: conv1's input gradient of a shape-changing block (BatchNorm backward of an fp32 forward tensor in its fill, backward
    sums against another in its epilogue) and the shortcut's in ONE launch = the two launches one after the other."""
    rng = np.random.default_rng(B + Cin + 1)
    c3 = T.Conv(L, 0.05 * rng.standard_normal((3, 3, Cin, Cout)), 2, "same"); c1 = T.Conv(L, 0.1 * rng.standard_normal((1, 1, Cin, Cout)), 2, "valid")
    ho, wo, _, _ = c3.geo(H, W)
    g = T.dbf(T.grid(rng, (B, ho, wo, Cout))); da = T.dbf(T.grid(rng, (B, ho, wo, Cout)))
    yb = f32(rng.standard_normal((B, ho, wo, Cout)) + 0.2); xin = f32(rng.standard_normal((B, H, W, Cin)))
    bn = T.BN(L, Cout, rng); bn.set_fwd_stats(yb); bn.set_saved(); bn.set_bwd_stats(T.val(da), yb)
    bn0 = T.BN(L, Cin, rng); bn0.set_fwd_stats(xin); bn0.set_saved()
    ybd, xind = T.dev(yb), T.dev(xin)

    def args3(dx, bins):
        _, _, pt_, pl_ = c3.geo(H, W)
        return (da, ybd, None, c3.wd, None, None, dx, None, B, ho, wo, Cout, H, W, Cin, 3, 1, 2, 2 - pt_, 2 - pl_, 2, bn.ref(bn.n, "bwd", False), 2, bins, xind, bn0.d["saved"])

    def args1(dx):
        return (g, None, None, c1.wd, None, None, dx, None, B, ho, wo, Cout, H, W, Cin, 1, 1, 2, 0, 0, 0, None, 0, None, None, None)

    mk = lambda: torch.full((B, H, W, Cin), 7.0, dtype=torch.float16, device="cuda")
    d3a, d1a, d3b, d1b = mk(), mk(), mk(), mk()
    ba, bb = torch.zeros_like(bn0.d["bbins"]), torch.zeros_like(bn0.d["bbins"])
    L.call("sar_conv2d_f16a", *args3(d3a, ba)); L.call("sar_conv2d_f16a", *args1(d1a))
    paired = L.conv_pair(L.ConvArgs.of(*args3(d3b, bb)), L.ConvArgs.of(*args1(d1b)), "f16a")
    torch.cuda.synchronize()
    assert paired
    assert torch.equal(d3a, d3b) and torch.equal(d1a, d1b)
    T.check(bb.sum(0), T.val(ba.sum(0)), 1e-5, "backward sums out of the paired launch")
    assert float(ba.abs().sum()) > 0
