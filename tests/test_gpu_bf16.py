"""GPU parity tests of the bf16 trunk kernels (csrc/conv_bf16.hip) against the CPU oracle (oracle/ops_np.py, float64).

Inputs are drawn on the bf16 grid (activations, gradients) or rounded to it exactly as the kernels do (weights, the
BatchNorm+ReLU / BatchNorm-backward transforms evaluated in the patch fill), so products are exact in fp32 and the only
differences left are fp32 accumulation order and the final bf16 rounding of stored tensors:
  stored bf16 tensors : |err| <= 4e-3 of the reference's max-abs (half an ulp is 2^-9 = 2e-3 relative)
  fp32 outputs        : weight gradients, statistics  <= 1e-3 / 1e-4 of max-abs
Every distinct conv shape of the bench configuration (res34, F=32, 400x80, B=64) is covered, fwd + dgrad + wgrad.
"""
import numpy as np
import pytest
import torch

from oracle import ops_np as O

pytestmark = pytest.mark.gpu
BF, SFX = torch.bfloat16, "bf16"
TOL_BF, TOL_F32 = 4e-3, 1e-3


X3 = False                                                           # the split-precision build: fp32 tensors in HBM, (hi, lo) bf16 operands in the kernels


@pytest.fixture(scope="module", autouse=True, params=["bf16", "fp16", "x3"])
def two_byte_type(request):
    """The module runs twice: against the bf16 build of the trunk kernels (entry points sar_*_bf16) and against the fp16 build
    (sar_*_f16: the same sources with IEEE half as the 2-byte type, csrc/trunk16.h).  Inputs are drawn on the grid of the type under
    test; a stored tensor is within half an ulp of the float64 oracle: 2^-9 (bf16) / 2^-12 (fp16) relative, bounded at twice that of
    the reference's max-abs."""
    global BF, SFX, TOL_BF, X3
    old = (BF, SFX, TOL_BF, X3)
    # "x3" (round 4, the package default): the split-precision build (entry points sar_*_x3) -- tensors are fp32 in HBM, an MFMA operand is
    # hi + lo of what the fill produced, 2^-17 relative; the conv forward / input-gradient tests run in it on every shape (its weight gradients
    # are the fp16 build's: those tests skip).
    BF, SFX, TOL_BF, X3 = {"fp16": (torch.float16, "f16", 5e-4, False), "bf16": (torch.bfloat16, "bf16", 4e-3, False),
                           "x3": (torch.bfloat16, "x3", 3e-5, True)}[request.param]
    yield request.param
    BF, SFX, TOL_BF, X3 = old


def bench_shape_only_bf16(shape):
    """The full-size (B = 64) shapes run once, in the bf16 build; the fp16 build -- the same code with another element type -- takes
    the small cases and every fourth bench shape."""
    if SFX == "f16" and shape not in SOME:
        pytest.skip("full-size shape list: bf16 and split-precision builds only")


@pytest.fixture(scope="module")
def L():
    from aesrc2020_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.lib()
    return _lib


def rbf(a):
    """numpy -> values rounded to the bf16 grid (float64); split precision: to the hi + lo grid."""
    t = torch.tensor(np.ascontiguousarray(a), dtype=torch.float32)
    hi = t.to(BF)
    if X3:
        return (hi.double() + (t - hi.float()).to(BF).double()).numpy()
    return hi.double().numpy()


def dbf(a):
    """-> the device tensor of the build under test (split precision: fp32)."""
    t = torch.tensor(np.ascontiguousarray(a), dtype=torch.float32)
    return t.cuda() if X3 else t.to(BF).cuda()


def empty16(*shape):
    return torch.empty(*shape, dtype=torch.float32 if X3 else BF, device="cuda")


def side16(shape):
    """A `side` tensor (the operand copy the fills write for the weight gradients): the build's 2-byte type; split precision: ONE fp16 tensor."""
    return torch.full(tuple(shape), float("nan"), dtype=torch.float16 if X3 else BF, device="cuda")


def x3_tail(side2=None, side_scale=1.0):
    """sar_conv2d_x3's two extra arguments (the fill's result as planes; the scale of the fp16 `side`)."""
    return (side2, float(side_scale)) if X3 else ()


def no_x3(what):
    if X3:
        pytest.skip("split precision: %s" % what)


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")


def val(t):
    return t.detach().float().cpu().double().numpy()


def check(got, ref, tol, name):
    got = val(got) if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.all(np.isfinite(got)), name
    err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-9)
    assert err <= tol, "%s: rel-to-max error %.3e > %.1e" % (name, err, tol)


class Conv:
    """Host mirror of one conv layer: packs the weight streams with the library and knows the TF geometry."""

    def __init__(self, L, w, stride, padding):
        self.L, self.w, self.s = L, np.asarray(w, np.float32), stride
        self.k, _, self.cin, self.cout = w.shape
        self.padding = padding
        taps = self.k * self.k
        npl = 2 if X3 else 1
        nf = (self.cout // 32) * (npl * taps * self.cin // 16 + 8) * 512
        nd = (self.cin // 32) * (npl * taps * self.cout // 16 + 8) * 512
        self.wdev = dev(self.w)
        self.wf = torch.zeros(nf, dtype=BF, device="cuda"); self.wd = torch.zeros(nd, dtype=BF, device="cuda")
        nb = -(-(nf + nd) // 8 // 256)
        tab = torch.tensor([[self.wdev.data_ptr(), self.wf.data_ptr(), self.wd.data_ptr(), taps, self.cin, self.cout, 0, 0]],
                           dtype=torch.int64, device="cuda")
        L.call("sar_conv_pack_weights_" + SFX, tab, 1, nb)
        self.wq = rbf(self.w)                                   # what the kernels multiply with

    def geo(self, H, W):
        if self.padding == "same":
            (ho, pt, _), (wo, pl, _) = O.same_pads(H, self.k, self.s), O.same_pads(W, self.k, self.s)
        else:
            ho, wo, pt, pl = O.valid_out(H, self.k, self.s), O.valid_out(W, self.k, self.s), 0, 0
        return ho, wo, pt, pl

    def fwd(self, x, bias=None, res=None, fill=0, fbn=None, e_bins=None, side=None, side2=None, side_scale=1.0):
        B, H, W, _ = x.shape[-4:]
        ho, wo, pt, pl = self.geo(H, W)
        y = empty16(B, ho, wo, self.cout)
        self.L.call("sar_conv2d_" + SFX, x, None, None, self.wf, bias, res, y, side, B, H, W, self.cin, ho, wo, self.cout, self.k, self.s, 1, pt, pl,
                    fill, fbn, 1 if e_bins is not None else 0, e_bins, None, None, *x3_tail(side2, side_scale))
        return y

    def dgrad(self, dy, H, W, dy2=None, fill=0, fbn=None, res=None, e_bins=None, e_y=None, e_saved=None, side=None, dy3=None, side2=None, side_scale=1.0):
        B, ho, wo, _ = dy.shape[-4:]
        _, _, pt, pl = self.geo(H, W)
        dx = empty16(B, H, W, self.cin)
        self.L.call("sar_conv2d_" + SFX, dy, dy2, dy3, self.wd, None, res, dx, side, B, ho, wo, self.cout, H, W, self.cin, self.k, 1, self.s,
                    self.k - 1 - pt, self.k - 1 - pl, fill, fbn, 2 if e_bins is not None else 0, e_bins, e_y, e_saved, *x3_tail(side2, side_scale))
        return dx

    def desc(self, x, dy, dw, db):
        B, H, W, _ = x.shape
        ho, wo, pt, pl = self.geo(H, W)
        return dict(x=x, dy=dy, dw=dw, dbias=db, B=B, H=H, W=W, Cin=self.cin, Ho=ho, Wo=wo, Cout=self.cout, K=self.k, s=self.s, pt=pt, pl=pl)

    def wgrad(self, x, dy, bias=True, target_blocks=768, tile=(1, 1)):
        """One-convolution plan, added to a non-zero gradient buffer (the kernel accumulates)."""
        assert not X3
        dw = torch.ones(self.k, self.k, self.cin, self.cout, device="cuda"); db = torch.zeros(self.cout, device="cuda")
        plan = self.L.WgradPlan([self.desc(x, dy, dw, db if bias else None)], target_blocks, tile, SFX)
        plan.launch()
        self.last_plan = plan
        return dw - 1.0, db


class BN:
    """Host mirror of one BatchNorm: device state + float64 reference of the constants the kernels derive."""
    EPS, MOM = 1e-3, 0.99

    def __init__(self, L, C, rng):
        self.L, self.C = L, C
        self.NB = L.bn_bins()
        self.gamma = rng.uniform(0.5, 1.5, C).astype(np.float32); self.beta = (0.3 * rng.standard_normal(C)).astype(np.float32)
        self.d = dict(gamma=dev(self.gamma), beta=dev(self.beta), bins=torch.zeros(self.NB, 2, C, device="cuda"),
                      bbins=torch.zeros(self.NB, 2, C, device="cuda"), saved=torch.zeros(4, C, device="cuda"),
                      mm=torch.zeros(C, device="cuda"), mv=torch.ones(C, device="cuda"), dg=torch.zeros(C, device="cuda"),
                      db=torch.zeros(C, device="cuda"))

    def ref(self, R, stats, write, ginv=1.0):
        r = self.L.BnRef()
        d = self.d
        r.gamma, r.beta, r.saved = d["gamma"].data_ptr(), d["beta"].data_ptr(), d["saved"].data_ptr()
        r.bins = d["bins"].data_ptr() if stats == "fwd" else None
        r.bbins = d["bbins"].data_ptr() if stats == "bwd" else None
        r.moving_mean, r.moving_var, r.dgamma, r.dbeta = d["mm"].data_ptr(), d["mv"].data_ptr(), d["dg"].data_ptr(), d["db"].data_ptr()
        r.inv_n, r.unbias, r.eps, r.momentum, r.write = 1.0 / R, R / max(R - 1.0, 1.0), self.EPS, self.MOM, int(write)
        r.ginv = float(ginv)
        return r

    def set_fwd_stats(self, x):
        """Fill the forward bins as a producer conv's epilogue would (split over two bins) and the f64 constants."""
        x2 = x.reshape(-1, self.C)
        s1, s2 = x2.sum(0), (x2 * x2).sum(0)
        bins = np.zeros((self.NB, 2, self.C), np.float32)
        bins[3, 0], bins[3, 1] = 0.25 * s1, 0.25 * s2
        bins[11, 0], bins[11, 1] = s1 - bins[3, 0].astype(np.float64), s2 - bins[3, 1].astype(np.float64)
        self.d["bins"].copy_(dev(bins))
        n = x2.shape[0]
        self.mean = s1 / n; self.var = np.maximum(s2 / n - self.mean ** 2, 0.0)
        self.rstd = 1.0 / np.sqrt(self.var + self.EPS)
        self.scale = self.gamma * self.rstd; self.shift = self.beta - self.mean * self.scale
        self.n = n

    def set_saved(self):
        self.d["saved"].copy_(dev(np.stack([self.mean, self.rstd, self.scale, self.shift])))

    def act(self, x):
        """relu(bn(x)) as the fill computes it, on the bf16 grid."""
        s, h = self.scale.astype(np.float32).astype(np.float64), self.shift.astype(np.float32).astype(np.float64)
        return rbf(np.maximum(x * s + h, 0.0))

    def bwd_sums(self, da, y):
        y2, d2 = y.reshape(-1, self.C), da.reshape(-1, self.C)
        s, h = self.scale.astype(np.float32).astype(np.float64), self.shift.astype(np.float32).astype(np.float64)
        dz = np.where(y2 * s + h > 0, d2, 0.0)
        xh = (y2 - self.mean.astype(np.float32)) * self.rstd.astype(np.float32)
        return dz.sum(0), (dz * xh).sum(0), dz, xh

    def set_bwd_stats(self, da, y):
        S1, S2, dz, xh = self.bwd_sums(da, y)
        bb = np.zeros((self.NB, 2, self.C), np.float32)
        bb[5, 0], bb[5, 1] = S1, S2
        self.d["bbins"].copy_(dev(bb))
        self.S1, self.S2 = S1, S2
        return dz, xh

    def dx(self, da, y):
        """BatchNorm+ReLU backward as the fills / apply pass compute it (float64 on the same constants)."""
        S1, S2, dz, xh = self.bwd_sums(da, y)
        n = dz.shape[0]
        out = self.scale.astype(np.float32).astype(np.float64) * (dz - np.float32(S1 / n) - xh * np.float32(S2 / n))
        return out.reshape(da.shape)


# every distinct 3x3 / 1x1 conv of the bench trunk (res34, F=32, T=400: pooled map 100x20x64) + small ragged cases
SHAPES = [
    # B, H, W, Cin, Cout, k, stride, padding
    (2, 9, 7, 32, 32, 3, 1, "same"),
    (3, 13, 3, 64, 64, 3, 1, "same"),                # several images per pixel tile
    (5, 13, 3, 256, 256, 3, 1, "same"),
    (2, 11, 20, 32, 64, 3, 2, "same"),               # odd height, stride 2
    (2, 10, 6, 64, 128, 1, 2, "valid"),
    (1, 4, 4, 512, 512, 3, 1, "same"),               # two channel halves per tap
    (6, 13, 3, 512, 512, 3, 1, "same"),              # res_filters = 64 (round 3): stage 3 on 32-pixel tiles (the 64-pixel patch of 512 channels exceeds the LDS)
    (4, 25, 5, 256, 512, 3, 2, "same"),              # ... its entry: forward from 256 channels, input gradient from 512
    (4, 25, 5, 256, 512, 1, 2, "valid"),
    (64, 100, 20, 64, 32, 3, 1, "same"),             # stage 0, block 0 conv1
    (64, 100, 20, 64, 32, 1, 1, "valid"),            # stage 0 shortcut
    (64, 100, 20, 32, 32, 3, 1, "same"),             # stage 0
    (64, 100, 20, 32, 64, 3, 2, "same"),             # stage 1 entry
    (64, 100, 20, 32, 64, 1, 2, "valid"),
    (64, 50, 10, 64, 64, 3, 1, "same"),              # stage 1
    (64, 50, 10, 64, 128, 3, 2, "same"),             # stage 2 entry
    (64, 50, 10, 64, 128, 1, 2, "valid"),
    (64, 25, 5, 128, 128, 3, 1, "same"),             # stage 2
    (64, 25, 5, 128, 256, 3, 2, "same"),             # stage 3 entry
    (64, 25, 5, 128, 256, 1, 2, "valid"),
    (64, 13, 3, 256, 256, 3, 1, "same"),             # stage 3
]
def _ids(shapes):
    return ["%dx%dx%dx%d-%d_k%d_s%d" % s[:7] for s in shapes]


IDS = _ids(SHAPES)
SOME = SHAPES[:9] + SHAPES[9::4]                  # the fill variants: all small cases + every fourth bench shape
IDS_SOME = _ids(SOME)


def grid(rng, shape, scale=1.0):
    return rbf(scale * rng.standard_normal(shape))


@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_conv_fwd_plain_residual_and_statistics(L, shape):
    bench_shape_only_bf16(shape)
    B, H, W, Cin, Cout, k, s, pad = shape
    rng = np.random.default_rng(sum(shape[:7]))
    x = grid(rng, (B, H, W, Cin)); w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)
    bias = rng.standard_normal(Cout).astype(np.float32)
    cv = Conv(L, w, s, pad)
    ref, _ = O.conv2d_fwd(x, cv.wq, bias, (s, s), pad)
    res = grid(rng, ref.shape)
    bins = torch.zeros(L.bn_bins(), 2, Cout, device="cuda")
    y = cv.fwd(dbf(x), dev(bias), dbf(res), e_bins=bins)
    check(y, ref + res, TOL_BF, "y")
    yv = val(y).reshape(-1, Cout)
    tot = val(bins).sum(0)
    check(tot[0], yv.sum(0), 1e-4, "sum(y)"); check(tot[1], (yv * yv).sum(0), 1e-4, "sum(y^2)")
    y2 = cv.fwd(dbf(x), None, None)
    check(y2, ref - bias, TOL_BF, "y (no bias / residual)")


@pytest.mark.parametrize("shape", SOME, ids=IDS_SOME)
def test_conv_fwd_with_batchnorm_relu_in_the_fill(L, shape):
    B, H, W, Cin, Cout, k, s, pad = shape
    rng = np.random.default_rng(1 + sum(shape[:7]))
    x = grid(rng, (B, H, W, Cin)) + rbf(0.5 * rng.standard_normal(Cin))
    x = rbf(x)
    w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)
    cv = Conv(L, w, s, pad); bn = BN(L, Cin, rng)
    bn.set_fwd_stats(x)
    ref, _ = O.conv2d_fwd(bn.act(x), cv.wq, None, (s, s), pad)
    side = side16((B, H, W, Cin))
    SC = 8.0 if X3 else 1.0                                       # split precision: side = fp16(relu(bn(x)) * side_scale)
    y = cv.fwd(dbf(x), None, None, fill=1, fbn=bn.ref(bn.n, "fwd", True), side=side, side_scale=SC)
    check(y, ref, TOL_BF, "y")
    if not (k == 1 and s == 2):                                   # a strided 1x1 conv never touches the odd pixels (not a case the trunk has)
        check(side.float() / SC, bn.act(x), 6e-4 if X3 else TOL_BF, "side = relu(bn(x)) (every pixel written exactly once)")
    check(bn.d["saved"], np.stack([bn.mean, bn.rstd, bn.scale, bn.shift]), 1e-5, "saved mean/rstd/scale/shift")
    check(bn.d["mm"], (1 - BN.MOM) * bn.mean, 1e-5, "moving mean")
    check(bn.d["mv"], BN.MOM + (1 - BN.MOM) * bn.var * bn.n / (bn.n - 1.0), 1e-5, "moving variance (Bessel)")
    # the same through the saved constants (backward consumers, inference): no further side effects
    y2 = cv.fwd(dbf(x), None, None, fill=1, fbn=bn.ref(bn.n, "saved", False))
    assert torch.equal(y, y2)
    check(bn.d["mm"], (1 - BN.MOM) * bn.mean, 1e-5, "moving mean untouched")


@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_conv_dgrad_plain_and_batchnorm_sums(L, shape):
    bench_shape_only_bf16(shape)
    B, H, W, Cin, Cout, k, s, pad = shape
    rng = np.random.default_rng(2 + sum(shape[:7]))
    w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cout)
    cv = Conv(L, w, s, pad)
    ho, wo, _, _ = cv.geo(H, W)
    x0 = np.zeros((B, H, W, Cin))
    _, cache = O.conv2d_fwd(x0, cv.wq, None, (s, s), pad)
    dy = grid(rng, (B, ho, wo, Cout))
    ref = O.conv2d_bwd(dy, cache)[0]
    res = grid(rng, ref.shape)
    dx = cv.dgrad(dbf(dy), H, W, res=dbf(res))
    check(dx, ref + res, TOL_BF, "dx + res")
    # epilogue sums of the BatchNorm in front of the conv (input xin), of the fp32 accumulators (before the bf16 store)
    bn = BN(L, Cin, rng)
    xin = rbf(grid(rng, (B, H, W, Cin)) + 0.3)
    bn.set_fwd_stats(xin); bn.set_saved()
    dx2 = cv.dgrad(dbf(dy), H, W, e_bins=bn.d["bbins"], e_y=dbf(xin), e_saved=bn.d["saved"])
    check(dx2, ref, TOL_BF, "dx")
    S1, S2, _, _ = bn.bwd_sums(ref, xin)
    tot = val(bn.d["bbins"]).sum(0)
    check(tot[0], S1, 2e-4, "sum dz"); check(tot[1], S2, 2e-4, "sum dz*xhat")


@pytest.mark.parametrize("shape", SOME, ids=IDS_SOME)
def test_conv_dgrad_with_batchnorm_backward_in_the_fill(L, shape):
    B, H, W, Cin, Cout, k, s, pad = shape
    rng = np.random.default_rng(3 + sum(shape[:7]))
    w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cout)
    cv = Conv(L, w, s, pad)
    ho, wo, _, _ = cv.geo(H, W)
    _, cache = O.conv2d_fwd(np.zeros((B, H, W, Cin)), cv.wq, None, (s, s), pad)
    da = grid(rng, (B, ho, wo, Cout)); yb = rbf(grid(rng, (B, ho, wo, Cout)) + 0.2)
    bn = BN(L, Cout, rng)
    bn.set_fwd_stats(yb); bn.set_saved(); bn.set_bwd_stats(da, yb)
    dyq = rbf(bn.dx(da, yb))                                    # what the fill writes to LDS
    ref = O.conv2d_bwd(dyq, cache)[0]
    side = side16(da.shape)
    SC = 64.0 if X3 else 1.0
    dx = cv.dgrad(dbf(da), H, W, dy2=dbf(yb), fill=2, fbn=bn.ref(bn.n, "bwd", True), side=side, side_scale=SC)
    check(dx, ref, TOL_BF, "dx")
    check(side.float() / SC, dyq, 6e-4 if X3 else TOL_BF, "side = BatchNorm backward of (da, y)")
    check(bn.d["dg"], bn.S2, 1e-5, "dgamma"); check(bn.d["db"], bn.S1, 1e-5, "dbeta")
    # fill 3: + the shortcut gradient -- the gradient a residual block hands to the block in front of it (resnet.py:96-125)
    add = grid(rng, da.shape)
    dyq3 = rbf(bn.dx(da, yb) + add)
    ref3 = O.conv2d_bwd(dyq3, cache)[0]
    side3 = side16(da.shape)
    side23 = empty16(*da.shape) if X3 else None                   # split precision: the materialised gradient as planes, too
    dx3 = cv.dgrad(dbf(da), H, W, dy2=dbf(yb), fill=3, fbn=bn.ref(bn.n, "bwd", False), side=side3, dy3=dbf(add), side2=side23, side_scale=SC)
    check(dx3, ref3, TOL_BF, "dx (fill 3)"); check(side3.float() / SC, dyq3, 6e-4 if X3 else TOL_BF, "side = BatchNorm backward + shortcut gradient")
    if X3:
        check(side23, dyq3, TOL_BF, "side2 = the same as planes")
    check(bn.d["dg"], bn.S2, 1e-5, "dgamma untouched by a launch that does not own the side effects")


@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_conv_wgrad(L, shape):
    no_x3("the weight gradients are the fp16 build's")
    bench_shape_only_bf16(shape)
    B, H, W, Cin, Cout, k, s, pad = shape
    rng = np.random.default_rng(4 + sum(shape[:7]))
    w = rng.standard_normal((k, k, Cin, Cout))
    cv = Conv(L, w, s, pad)
    ho, wo, _, _ = cv.geo(H, W)
    x = rbf(grid(rng, (B, H, W, Cin)) + 0.4); dy = grid(rng, (B, ho, wo, Cout))
    _, cache = O.conv2d_fwd(x, cv.wq, np.zeros(Cout), (s, s), pad)
    _, dw_ref, db_ref = O.conv2d_bwd(dy, cache)
    dw, db = cv.wgrad(dbf(x), dbf(dy))
    check(dw, dw_ref, TOL_F32, "dw"); check(db, db_ref, TOL_F32, "db")
    dw2, db2 = cv.wgrad(dbf(x), dbf(dy), bias=False, target_blocks=1)           # one item per tile: the read-add-write flush
    check(dw2, dw_ref, TOL_F32, "dw (no bias gradient, single slice)")
    assert float(db2.abs().max()) == 0.0
    items = np.frombuffer(cv.last_plan.items.cpu().numpy().tobytes(), np.int32).reshape(-1, 8)
    # (3x3: a run is also bounded by the item's source-offset table, so the largest shapes still get several slices)
    assert bool(items[:, 5].all()) == (len(items) == (Cin // 32) * (Cout // 32)) and (k == 3 or items[:, 5].all())
    if k == 3 and s == 1:                                                         # the wider channel tiles (stride-1 convs of the trunk)
        for tile in ((2, 2), (1, 2), (2, 1)):
            if Cin >= 32 * tile[0] and Cout >= 32 * tile[1]:
                dwt, dbt = cv.wgrad(dbf(x), dbf(dy), tile=tile, target_blocks=300)
                check(dwt, dw_ref, TOL_F32, "dw, tile %dx%d" % tile); check(dbt, db_ref, TOL_F32, "db, tile %dx%d" % tile)
                assert cv.last_plan.np in (512, 1024) and 2 <= cv.last_plan.nbuf <= 4
    dw3, db3 = cv.wgrad(dbf(x), dbf(dy), target_blocks=100000)                   # shortest runs: the atomic flush
    check(dw3, dw_ref, TOL_F32, "dw (shortest runs)"); check(db3, db_ref, TOL_F32, "db (shortest runs)")
    items = np.frombuffer(cv.last_plan.items.cpu().numpy().tobytes(), np.int32).reshape(-1, 8)
    ch = 128 if (cv.last_plan.launch4[3] >> 8) & 1 else 64                       # pixels per chunk (round 6: 128 where the convolution fits the ring with one in flight)
    nch = -(-B * ho * wo // ch)
    assert items[:, 4].max() * ch <= (16 if k == 3 else 8) * 64 and (nch <= items[:, 4].max() or not items[:, 5].any())   # the planner's shortest runs


def test_conv_wgrad_many_convolutions_in_one_launch(L):
    """The train step's form: the 3x3 stride-1 convs of several stages in one plan, the 1x1 convs in another."""
    no_x3("the weight gradients are the fp16 build's")
    rng = np.random.default_rng(99)
    for k, tile, shapes in ((3, (1, 1), [(4, 25, 20, 32, 32, 1), (4, 13, 10, 64, 64, 1), (4, 7, 5, 128, 128, 1), (4, 4, 3, 256, 256, 1), (4, 25, 20, 64, 32, 1)]),
                            (3, (2, 2), [(4, 13, 10, 64, 64, 1), (4, 7, 5, 128, 128, 1), (4, 4, 3, 256, 256, 1), (5, 13, 3, 256, 256, 1)]),
                            (3, (1, 1), [(3, 25, 20, 32, 64, 2), (3, 13, 10, 64, 128, 2)]),
                            (1, (1, 1), [(4, 25, 20, 64, 32, 1), (4, 25, 20, 32, 64, 2), (4, 7, 5, 128, 256, 2)])):
        descs, refs, outs = [], [], []
        for (B, H, W, Cin, Cout, s) in shapes:
            pad = "same" if k == 3 else "valid"
            cv = Conv(L, rng.standard_normal((k, k, Cin, Cout)), s, pad)
            ho, wo, _, _ = cv.geo(H, W)
            x = rbf(grid(rng, (B, H, W, Cin)) + 0.4); dy = grid(rng, (B, ho, wo, Cout))
            _, cache = O.conv2d_fwd(x, cv.wq, np.zeros(Cout), (s, s), pad)
            _, dw_ref, db_ref = O.conv2d_bwd(dy, cache)
            dw = torch.zeros(k, k, Cin, Cout, device="cuda"); db = torch.zeros(Cout, device="cuda")
            descs.append(cv.desc(dbf(x), dbf(dy), dw, db)); refs.append((dw_ref, db_ref)); outs.append((dw, db))
        plan = L.WgradPlan(descs, 64, tile, SFX)
        plan.launch()
        for i, ((dw, db), (dw_ref, db_ref)) in enumerate(zip(outs, refs)):
            check(dw, dw_ref, TOL_F32, "dw of conv %d (k=%d)" % (i, k)); check(db, db_ref, TOL_F32, "db of conv %d (k=%d)" % (i, k))
        items = np.frombuffer(plan.items.cpu().numpy().tobytes(), np.int32).reshape(-1, 8)
        assert (np.diff(items[:, 4]) <= 0).all(), "longest items first"


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(64, 50, 10, 64, 128), (64, 25, 5, 128, 256), (64, 100, 20, 32, 64), (5, 9, 7, 64, 128), (4, 25, 5, 256, 512)])
def test_conv_and_its_shortcut_in_one_launch(L, B, H, W, Cin, Cout):
    """sar_conv2d_pair_*: the stride-2 3x3 conv of a shape-changing block (BatchNorm + ReLU in its fill, statistics out) and the block's
    1x1 shortcut in ONE launch give exactly what the two launches give one after the other; the same for their input gradients.  The last
    shape (256 -> 512: the 3x3 runs on 32-pixel tiles, the 1x1 on 64) does not pair up in the single-plane builds: the call says so and launches nothing."""
    rng = np.random.default_rng(B + Cin)
    c3 = Conv(L, 0.05 * rng.standard_normal((3, 3, Cin, Cout)), 2, "same"); c1 = Conv(L, 0.1 * rng.standard_normal((1, 1, Cin, Cout)), 2, "valid")
    x = dbf(grid(rng, (B, H, W, Cin)))
    ho, wo, pt, pl = c3.geo(H, W); assert c1.geo(H, W)[:2] == (ho, wo)
    bias3, bias1 = dev(rng.standard_normal(Cout)), dev(rng.standard_normal(Cout))

    def fwd_args(cv, y, bias, bins):
        _, _, pt_, pl_ = cv.geo(H, W)
        return (x, None, None, cv.wf, bias, None, y, None, B, H, W, Cin, ho, wo, Cout, cv.k, 2, 1, pt_, pl_, 0, None, 1 if bins is not None else 0, bins, None, None) + x3_tail()

    mk = lambda: torch.full((B, ho, wo, Cout), 7.0, dtype=torch.float32 if X3 else BF, device="cuda")
    y3a, y1a, y3b, y1b = mk(), mk(), mk(), mk()
    binsa, binsb = torch.zeros(L.bn_bins(), 2, Cout, device="cuda"), torch.zeros(L.bn_bins(), 2, Cout, device="cuda")
    L.call("sar_conv2d_" + SFX, *fwd_args(c3, y3a, bias3, binsa)); L.call("sar_conv2d_" + SFX, *fwd_args(c1, y1a, bias1, None))
    paired = L.conv_pair(L.ConvArgs.of(*fwd_args(c3, y3b, bias3, binsb)), L.ConvArgs.of(*fwd_args(c1, y1b, bias1, None)), SFX)
    torch.cuda.synchronize()
    # (which shapes do not pair up: the single-plane builds run the 3x3 from 256 channels on 32-pixel tiles and its 1x1 on 64; in split precision,
    # twice the patch, that happens from 128 channels -- the one mixed pair that has an instantiation since round 6 (the forward entry of the last stage at
    # res_filters 32) -- and from 256 both run the 32-pixel form and pair)
    if Cin == 256 and not X3:
        assert not paired and float((y3b.float() - 7.0).abs().max()) == 0.0      # nothing was launched
        return
    assert paired
    assert torch.equal(y3a, y3b) and torch.equal(y1a, y1b)
    check(binsb.sum(0), val(binsa.sum(0)), 1e-5, "statistics out of the paired launch")
    # input gradients: source = the block's output gradient (Cout channels), stride-2 convs as stride-1 convs over the dilated gradient
    g = dbf(grid(rng, (B, ho, wo, Cout)))

    def dgrad_args(cv, dx):
        _, _, pt_, pl_ = cv.geo(H, W)
        return (g, None, None, cv.wd, None, None, dx, None, B, ho, wo, Cout, H, W, Cin, cv.k, 1, 2, cv.k - 1 - pt_, cv.k - 1 - pl_, 0, None, 0, None, None, None) + x3_tail()

    mkx = lambda: torch.full((B, H, W, Cin), 7.0, dtype=torch.float32 if X3 else BF, device="cuda")
    d3a, d1a, d3b, d1b = mkx(), mkx(), mkx(), mkx()
    L.call("sar_conv2d_" + SFX, *dgrad_args(c3, d3a)); L.call("sar_conv2d_" + SFX, *dgrad_args(c1, d1a))
    paired = L.conv_pair(L.ConvArgs.of(*dgrad_args(c3, d3b)), L.ConvArgs.of(*dgrad_args(c1, d1b)), SFX)
    torch.cuda.synchronize()
    if paired:                                                                     # (32 output channels: the 3x3 and the 1x1 run on different block sizes)
        assert torch.equal(d3a, d3b) and torch.equal(d1a, d1b)
    else:
        # (... and in split precision the gradients from 512 channels: the 3x3 owns sixteen pixels of its 32-pixel tile, the 1x1 all of them)
        assert (Cin <= 32 or (X3 and Cin == 256)) and float((d3b.float() - 7.0).abs().max()) == 0.0


def test_wgrad_reductions_of_several_plans_in_one_launch(L):
    """sar_wgrad_reduce_multi_*: three plans of different tile classes (64-wide 3x3, 32-wide 3x3, 1x1) run with their reductions deferred
    (launch4[7] = 1), then ONE launch sums the partial tiles of all of them -- same gradients as each plan reducing for itself."""
    no_x3("the weight gradients are the fp16 build's")
    rng = np.random.default_rng(77)
    cases = [((16, 50, 10, 64, 64, 3, 1, "same"), (2, 2)), ((16, 50, 20, 32, 32, 3, 1, "same"), (1, 1)), ((16, 50, 10, 64, 128, 1, 2, "valid"), (1, 1))]   # >= 32 chunks of 64 pixels each: the planner cuts runs of >= 16
    plans, outs = [], []
    for shape, tile in cases:
        B, H, W, Cin, Cout, k, s, pad = shape
        cv = Conv(L, rng.standard_normal((k, k, Cin, Cout)), s, pad)
        ho, wo, _, _ = cv.geo(H, W)
        x = rbf(grid(rng, (B, H, W, Cin)) + 0.4); dy = grid(rng, (B, ho, wo, Cout))
        _, cache = O.conv2d_fwd(x, cv.wq, np.zeros(Cout), (s, s), pad)
        _, dw_ref, db_ref = O.conv2d_bwd(dy, cache)
        dw = torch.ones(k, k, Cin, Cout, device="cuda"); db = torch.zeros(Cout, device="cuda")
        plan = L.WgradPlan([cv.desc(dbf(x), dbf(dy), dw, db)], 48, tile, SFX)       # 48 items: every gradient tile is shared by several
        assert plan.n_tiles > 0 and plan.n_slabs > plan.n_tiles
        plans.append(plan); outs.append((dw, db, dw_ref, db_ref))
    for p_ in plans:
        p_.launch(defer_reduce=True)
    for dw, _, _, _ in outs:                                                          # nothing has reached the gradients yet
        assert float((dw - 1.0).abs().max()) == 0.0
    L.WgradPlan.reduce_multi(plans, SFX)
    for dw, db, dw_ref, db_ref in outs:
        check(dw - 1.0, dw_ref, TOL_F32, "dw (multi-plan reduction)"); check(db, db_ref, TOL_F32, "db")


@pytest.mark.parametrize("R,C", [(2496, 256), (77, 32), (128000, 32), (8000, 128), (5, 512)])
def test_batchnorm_passes(L, R, C):
    rng = np.random.default_rng(R + C)
    x = rbf(grid(rng, (R, C)) + 0.3 * rng.standard_normal(C)); da = grid(rng, (R, C)); add = grid(rng, (R, C))
    bn = BN(L, C, rng)
    bn.set_fwd_stats(x)
    out = torch.empty(R, C, device="cuda")
    L.call("sar_bn_apply_%s_f32" % SFX, dbf(x), bn.ref(R, "fwd", True), out, R, C, 1)
    s, h = bn.scale.astype(np.float32).astype(np.float64), bn.shift.astype(np.float32).astype(np.float64)
    check(out, np.maximum(x * s + h, 0), 1e-5, "final bn+relu")
    check(bn.d["saved"], np.stack([bn.mean, bn.rstd, bn.scale, bn.shift]), 1e-5, "saved")
    for f32 in (1, 0):
        bn.d["bbins"].zero_(); bn.d["dg"].zero_(); bn.d["db"].zero_()
        dd = dev(da) if f32 else dbf(da)
        L.call("sar_bn_bwd_reduce_" + SFX, dd, f32, dbf(x), bn.d["saved"], bn.d["bbins"], R, C, 1)
        S1, S2, _, _ = bn.bwd_sums(da, x)
        tot = val(bn.d["bbins"]).sum(0)
        check(tot[0], S1, 2e-4, "S1"); check(tot[1], S2, 2e-4, "S2")
        g = empty16(R, C)
        if X3:                                                       # + the fp16 copy of the result times out_scale (the weight gradients' operand)
            g16 = side16((R, C))
            L.call("sar_bn_bwd_apply_x3", dd, f32, dbf(x), bn.ref(R, "bwd", True), dbf(add), g, g16, R, C, 1, 16.0)
            check(g16.float() / 16.0, bn.dx(da, x) + add, 6e-4, "bn bwd apply, fp16 copy")
        else:
            L.call("sar_bn_bwd_apply_" + SFX, dd, f32, dbf(x), bn.ref(R, "bwd", True), dbf(add), g, R, C, 1, 1.0)
        check(g, bn.dx(da, x) + add, TOL_BF, "bn bwd apply")
        check(bn.d["dg"], S2, 2e-4, "dgamma"); check(bn.d["db"], S1, 2e-4, "dbeta")


def test_gradient_scale_enters_at_the_final_batchnorm_and_leaves_with_the_parameter_gradients(L):
    """The fp16 trunk's static gradient scale S (include/sarnet.h, "fp16 trunk"; the entry points of both builds take it): out_scale
    multiplies the gradient where it enters, sarBnRef.ginv / sarWgradConv.ginv = 1/S divide the parameter gradients where they leave."""
    no_x3("the gradient stream is in true units; only the fp16 copies carry the scale (test_batchnorm_passes, the fill tests)")
    rng = np.random.default_rng(5)
    S = 256.0
    R, C = 640, 64
    x = rbf(grid(rng, (R, C)) + 0.3); da = grid(rng, (R, C))
    bn = BN(L, C, rng)
    bn.set_fwd_stats(x); bn.set_saved()
    L.call("sar_bn_bwd_reduce_" + SFX, dev(da), 1, dbf(x), bn.d["saved"], bn.d["bbins"], R, C, 1)
    g = torch.empty(R, C, dtype=BF, device="cuda")
    L.call("sar_bn_bwd_apply_" + SFX, dev(da), 1, dbf(x), bn.ref(R, "bwd", True), None, g, R, C, 1, S)
    check(g, S * bn.dx(da, x), TOL_BF, "scaled gradient")
    S1, S2, _, _ = bn.bwd_sums(da, x)
    check(bn.d["dg"], S2, 2e-4, "dgamma of the final BatchNorm: unscaled sums, ginv = 1")
    # a conv whose gradient operands carry S: the BatchNorm backward in its fill (sums scaled) and its weight gradient
    B, H, W, Cin, Cout = 3, 13, 5, 64, 64
    cv = Conv(L, rng.standard_normal((3, 3, Cin, Cout)) / 24.0, 1, "same")
    _, cache = O.conv2d_fwd(np.zeros((B, H, W, Cin)), cv.wq, None, (1, 1), "same")
    da2 = grid(rng, (B, H, W, Cout)); yb = rbf(grid(rng, (B, H, W, Cout)) + 0.2)
    bn2 = BN(L, Cout, rng)
    bn2.set_fwd_stats(yb); bn2.set_saved(); bn2.set_bwd_stats(S * da2, yb)
    ref_dx = O.conv2d_bwd(rbf(S * bn2.dx(da2, yb)), cache)[0]
    dx = cv.dgrad(dbf(S * da2), H, W, dy2=dbf(yb), fill=2, fbn=bn2.ref(bn2.n, "bwd", True, 1.0 / S))
    check(dx, ref_dx, TOL_BF, "dx (scaled)")
    check(bn2.d["dg"], bn2.S2 / S, 1e-5, "dgamma = scaled sums * ginv"); check(bn2.d["db"], bn2.S1 / S, 1e-5, "dbeta")
    xw = rbf(grid(rng, (B, H, W, Cin)) + 0.4); dy = grid(rng, (B, H, W, Cout))
    _, cache = O.conv2d_fwd(xw, cv.wq, np.zeros(Cout), (1, 1), "same")
    _, dw_ref, db_ref = O.conv2d_bwd(dy, cache)
    dw = torch.zeros(3, 3, Cin, Cout, device="cuda"); db = torch.zeros(Cout, device="cuda")
    L.WgradPlan([dict(cv.desc(dbf(xw), dbf(S * dy), dw, db), ginv=1.0 / S)], 8, (2, 2), SFX).launch()
    check(dw, dw_ref, TOL_F32, "dw"); check(db, db_ref, TOL_F32, "db")
    cv1 = Conv(L, rng.standard_normal((1, 1, Cin, Cout)), 1, "valid")
    _, cache = O.conv2d_fwd(xw, cv1.wq, np.zeros(Cout), (1, 1), "valid")
    _, dw_ref, db_ref = O.conv2d_bwd(dy, cache)
    dw1 = torch.zeros(1, 1, Cin, Cout, device="cuda"); db1 = torch.zeros(Cout, device="cuda")
    L.WgradPlan([dict(cv1.desc(dbf(xw), dbf(S * dy), dw1, db1), ginv=1.0 / S)], 8, (1, 1), SFX).launch()
    check(dw1, dw_ref, TOL_F32, "dw (1x1)"); check(db1, db_ref, TOL_F32, "db (1x1)")


def test_casts(L):
    rng = np.random.default_rng(0)
    a = rng.standard_normal(8 * 1000 + 8).astype(np.float32)
    if X3:                                                           # the trunk's format is fp32: a copy + (nullable) the fp16 copy times a scale
        y = empty16(a.size); y16 = side16((a.size,))
        L.call("sar_cast_f32_x3", dev(a), y, y16, 4.0, a.size)
        assert torch.equal(y.cpu(), torch.tensor(a))
        assert torch.equal(y16.cpu(), (torch.tensor(a) * 4.0).to(torch.float16))
        z = torch.empty(a.size, device="cuda")
        L.call("sar_cast_x3_f32", y, z, a.size)
        assert torch.equal(z.cpu(), torch.tensor(a))
        return
    y = torch.empty(a.size, dtype=BF, device="cuda")
    L.call("sar_cast_f32_" + SFX, dev(a), y, a.size)
    assert torch.equal(y.cpu(), torch.tensor(a).to(BF))
    z = torch.empty(a.size, device="cuda")
    L.call("sar_cast_%s_f32" % SFX, y, z, a.size)
    assert torch.equal(z.cpu(), torch.tensor(a).to(BF).float())
