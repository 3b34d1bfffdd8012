"""Block-by-block parity of the bf16 trunk (resnet.py _forward16 / _backward16) against the CPU oracle.

Every stored tensor of the bf16 trunk is on the bf16 grid.  Two runs that differ in ONE last bit (fp32 vs float64 accumulation
order) decorrelate within a few layers -- a 2^-9 input perturbation moves a quarter of the next layer's outputs across a
rounding boundary -- so an end-to-end "rounding-aware" reference only agrees to the bf16 noise level (measured: 2 % on the
trunk output, 10-20 % on gradients, tests/test_gpu_bf16_model.py).  Wiring errors are found one block at a time instead: the
reference takes the tensors the HIP path itself stored as the block's INPUTS (block input x, conv1 output c1, incoming
gradient g -- tapped through ResNetTrunk.debug_tap before the scratch buffers are reused) and recomputes that block's
outputs with the float64 oracle (oracle/ops_np.py conv2d / BatchNorm algebra, resnet.py:96-125 wiring), rounding to bf16
exactly where the kernels store a tensor or write a fill operand to LDS.  One block deep, a wrong BatchNorm in a fill, a
missing shortcut gradient or a wrong stride-2 padding is an O(1) error; correct wiring leaves isolated last-bit flips.
Bounds: 1e-2 of each tensor's max-abs (parameter gradients: 5e-3).

Round 4: the same walk in split precision (trunk dtype "bf16x3", csrc/trunk16.h SAR_X3): every stored tensor is a (hi, lo) pair of bf16
planes, the rounding function is x -> bf16(x) + bf16(x - bf16(x)) (16 mantissa bits), and the bounds are 20x tighter (5e-4; parameter
gradients keep 5e-3: their operands are single fp16 copies).

Round 5: the split-precision mode's MIXED BACKWARD (ResNetTrunk.bwd16, sar_conv2d_f16a): forward tensors and operands as above; every gradient
tensor of the residual stream is fp16 times the static scale (rounding function x -> fp16(x * S) / S), held to 2e-3 (half an ulp of fp16 is
2.4e-4 relative; the input-gradient convs also round their kernels to fp16, 2^-12 per product, which the reference below does not model).
"""
import numpy as np
import pytest
import torch

from oracle import ops_np as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
EPS = 1e-3


X3 = False                                                           # set per test: the split-precision walk


def rbf(a):
    """Round to what a stored tensor / an MFMA operand holds: the bf16 grid; split precision: hi + lo planes."""
    t = torch.tensor(np.ascontiguousarray(a), dtype=torch.float32)
    hi = t.to(BF).float()
    if X3:
        hi = hi + (t - hi).to(BF).float()
    return hi.double().numpy()


GS = 0.0                                                             # > 0: the gradient stream is fp16 under this scale (the mixed backward)


def rg(a):
    """Round to what a stored GRADIENT tensor holds."""
    if GS > 0:
        return (torch.tensor(np.ascontiguousarray(a) * GS, dtype=torch.float32).to(torch.float16).double() / GS).numpy()
    return rbf(a)


def f32(a):
    return np.asarray(a, np.float32).astype(np.float64)


class RefBN:
    """Constants exactly as bn_prologue derives them from the statistics of the STORED (bf16) tensor."""

    def __init__(self, x, gamma, beta, saved=None):
        x2 = x.reshape(-1, x.shape[-1])
        self.n = x2.shape[0]
        self.mean = f32(x2.mean(0)); var = np.maximum((x2 * x2).mean(0) - x2.mean(0) ** 2, 0.0)
        self.rstd = f32(1.0 / np.sqrt(f32(var) + EPS))
        self.scale = f32(gamma * self.rstd); self.shift = f32(beta - self.mean * self.scale)
        self.x = x
        # The ReLU decision of the backward, z > 0 with z = fmaf(y, scale, shift), is taken with the DEVICE's constants (saved: mean, rstd, scale,
        # shift -- fp32 sums of atomics, 1e-7 off the float64 ones): an element within 1e-7 of zero otherwise flips, once in ~20 runs of this
        # test, and on the 84-pixel maps of the last stage ONE flipped element is percents of a dbeta.  The float64 expression below has the
        # sign of the device's single-rounded fmaf.  (What is checked about the constants themselves: that they agree to 1e-5.)
        self.msc, self.msh = self.scale, self.shift
        if saved is not None:
            C = x.shape[-1]
            dev = np.asarray(saved, np.float64).reshape(-1)[:4 * C].reshape(4, C)
            assert np.abs(dev[2] - self.scale).max() <= 1e-5 * np.abs(self.scale).max() + 1e-7, "BatchNorm scale"
            assert np.abs(dev[3] - self.shift).max() <= 1e-5 * np.abs(self.shift).max() + 1e-6, "BatchNorm shift"
            self.msc, self.msh = dev[2], dev[3]

    def act(self):
        return rbf(np.maximum(self.x * self.scale + self.shift, 0.0))

    def bwd(self, da, da_acc=None):
        """-> (dx unrounded, dgamma, dbeta): BatchNorm+ReLU backward with the kernel's operation order.  da: the STORED (bf16)
        gradient the consumer fill reads; da_acc: the values the two sums are taken over -- the producing conv's fp32
        accumulators, before the bf16 store (conv_bf16.hip epilogue, epi == 2)."""
        mask = self.x * self.msc + self.msh > 0
        dz = np.where(mask, da, 0.0)
        xh = (self.x - self.mean) * self.rstd
        ax = tuple(range(da.ndim - 1))
        dzs = dz if da_acc is None else np.where(mask, da_acc, 0.0)
        S1, S2 = dzs.sum(ax), (dzs * xh).sum(ax)
        dx = self.scale * (dz - f32(S1 / self.n) - xh * f32(S2 / self.n))
        return dx, S2, S1


def relerr(got, ref):
    got = got.detach().float().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
@pytest.mark.parametrize("res_type,B,T", [("res34", 3, 72), ("res18", 2, 40), ("res34", 4, 200)])
def test_bf16_trunk_block_by_block_against_oracle(res_type, B, T, mode):
    from aesrc2020_amd import _lib
    from aesrc2020_amd.engine import ParamStore, store_scope
    from aesrc2020_amd.resnet import ResNetTrunk
    global X3, GS
    X3 = mode == "bf16x3"
    ta, tg = (5e-4, 5e-3) if X3 else (1e-2, 5e-3)                  # bounds: stored activations / gradients of the stream, parameter gradients
    old = _lib.get_trunk_dtype()
    _lib.set_trunk_dtype(mode)
    try:
        store = ParamStore(seed=3)
        with store_scope(store):
            tr = ResNetTrunk(res_type, 32, (T, 80))
        store.finalize()
    finally:
        _lib.set_trunk_dtype(old)
    rng = np.random.default_rng(11)
    for r in store.refs:                                            # non-trivial BatchNorm parameters and biases
        if r.name.endswith("/gamma"):
            r.w.copy_(torch.tensor(rng.uniform(0.6, 1.4, r.shape), dtype=torch.float32))
        elif r.name.endswith("/beta") or r.name.endswith("/bias"):
            r.w.copy_(torch.tensor(0.2 * rng.standard_normal(r.shape), dtype=torch.float32))
    x = torch.tensor(rng.uniform(0, 1, (B, T, 80, 1)), dtype=torch.float32, device="cuda")
    store.zero_grads()
    out = tr.forward(x, B)
    dy_np = f32(rng.standard_normal(tuple(out.shape)))
    taps = {}
    tr.debug_tap = lambda n, g, da2, gin: taps.__setitem__(n, (g.clone(), da2.clone(), gin.clone()))
    g_final = None
    tr.backward(torch.tensor(dy_np, dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    GS = float(tr.gscale) if getattr(tr, "bwd16", False) else 0.0   # the mixed backward: gradient taps are fp16 tensors times the static scale
    tgs = 2e-3 if GS > 0 else ta                                    # bound of a stored gradient tensor of the stream
    def gval(t):
        return val(t) / GS if GS > 0 else val(t)
    P = {r.name: r.w.detach().cpu().double().numpy() for r in store.refs}
    G = {r.name: r.g.detach().cpu().double().numpy() for r in store.refs}
    def val(t):                                                     # (split precision: the trunk's tensors are fp32)
        return t.detach().float().cpu().double().numpy()
    errs = {}

    def chk(name, got, ref, tol):
        e = relerr(got, ref) if not isinstance(got, np.ndarray) else float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))
        errs[name] = e
        assert e < tol, (name, e)

    nblk = len(tr.blocks)
    cache = []
    # ---- forward, one block at a time from the stored inputs
    for n, blk in enumerate(tr.blocks):
        c1v, c2v, scv = blk["conv1"], blk["conv2"], blk["sc"]
        xin, c1 = val(blk["x"]), val(blk["c1"])
        bn1 = RefBN(xin, P[blk["bn1"].gamma.name], P[blk["bn1"].beta.name], blk["bn1"].saved.cpu().numpy()) if blk["bn1"] is not None else None
        a = bn1.act() if bn1 is not None else xin
        y1, cache1 = O.conv2d_fwd(a, rbf(P[c1v.kernel.name]), P[c1v.bias.name], (c1v.stride,) * 2, c1v.padding)
        chk("b%d c1" % n, c1, rbf(y1), ta)
        bn2 = RefBN(c1, P[blk["bn2"].gamma.name], P[blk["bn2"].beta.name], blk["bn2"].saved.cpu().numpy())
        y2, cache2 = O.conv2d_fwd(bn2.act(), rbf(P[c2v.kernel.name]), P[c2v.bias.name], (1, 1), "same")
        if scv is not None:
            ys, caches = O.conv2d_fwd(xin, rbf(P[scv.kernel.name]), P[scv.bias.name], (scv.stride,) * 2, scv.padding)
            sc = rbf(ys)
        else:
            sc, caches = xin, None
        ynext = val(tr.blocks[n + 1]["x"]) if n + 1 < nblk else val(tr.pre_final)
        chk("b%d y" % n, ynext, rbf(y2 + sc), ta)
        cache.append((bn1, bn2, cache1, cache2, caches))
    fb = RefBN(val(tr.pre_final), P[tr.final_bn.gamma.name], P[tr.final_bn.beta.name], tr.final_bn.saved.cpu().numpy())
    chk("trunk out", val(out), np.maximum(fb.x * fb.scale + fb.shift, 0.0), 1e-4)
    # ---- backward: final BatchNorm, then every block from its tapped incoming gradient
    dx, dgam, dbet = fb.bwd(dy_np)
    chk("final bn -> g", gval(taps[nblk - 1][0]), rg(dx), tgs)
    chk("final bn dgamma", G[tr.final_bn.gamma.name], dgam, 5e-3); chk("final bn dbeta", G[tr.final_bn.beta.name], dbet, 5e-3)
    for n in range(nblk - 1, -1, -1):
        blk = tr.blocks[n]
        bn1, bn2, cache1, cache2, caches = cache[n]
        c1v, c2v, scv = blk["conv1"], blk["conv2"], blk["sc"]
        g, da2_dev, gin_dev = (gval(t) for t in taps[n])
        if n + 1 < nblk:                                            # the chain: this block's g is the next block's gin
            assert torch.equal(taps[n][0], taps[n + 1][2])
        da2, dw2, db2 = O.conv2d_bwd(g, cache2)
        chk("b%d da2" % n, da2_dev, rg(da2), tgs)
        chk("b%d conv2 dw" % n, G[c2v.kernel.name], dw2, 5e-3)
        tb = 1e-3 if X3 else 1e-4                                   # (split precision: column sums of the fp16 copy of g, 2^-12 per element)
        assert np.abs(G[c2v.bias.name] - db2).max() < tb * np.abs(g).reshape(-1, g.shape[-1]).sum(0).max() + 1e-6, "b%d conv2 db" % n
        dc1, dg2, dbt2 = bn2.bwd(da2_dev, da2)                       # the STORED da2, as the fills read it; sums over the accumulators
        chk("b%d bn2 dgamma" % n, G[blk["bn2"].gamma.name], dg2, 5e-3); chk("b%d bn2 dbeta" % n, G[blk["bn2"].beta.name], dbt2, 5e-3)
        dc1 = rg(dc1)
        da1, dw1, _ = O.conv2d_bwd(dc1, cache1)
        chk("b%d conv1 dw" % n, G[c1v.kernel.name], dw1, 5e-3)
        assert np.abs(G[c1v.bias.name]).max() == 0.0                 # feeds a BatchNorm: exactly zero, not computed
        if scv is not None:
            dsc, dws, dbs = O.conv2d_bwd(g, caches)
            chk("b%d shortcut dw" % n, G[scv.kernel.name], dws, 5e-3)
            assert np.abs(G[scv.bias.name] - dbs).max() < tb * np.abs(g).reshape(-1, g.shape[-1]).sum(0).max() + 1e-6
            dsc = rg(dsc)
        else:
            dsc = g
        if bn1 is not None:
            dxb, dg1, dbt1 = bn1.bwd(rg(da1), da1)
            chk("b%d gin" % n, gin_dev, rg(dxb + dsc), tgs)
            chk("b%d bn1 dgamma" % n, G[blk["bn1"].gamma.name], dg1, 1e-2); chk("b%d bn1 dbeta" % n, G[blk["bn1"].beta.name], dbt1, 1e-2)
        else:
            chk("b%d gin" % n, gin_dev, rg(da1 + dsc), tgs)
    # ---- the stem (csrc/stem_bf16.hip: conv recomputed in every pass, no full-resolution map): wiring of its four launches
    xq, wq = rbf(val(x)), rbf(P[tr.stem.kernel.name])
    ys, ccache = O.conv2d_fwd(xq, wq, P[tr.stem.bias.name], (2, 2), "same")
    zs, bcache = O.batchnorm_fwd(ys, P[tr.stem_bn.gamma.name], P[tr.stem_bn.beta.name], 1e-3)
    as_ = O.relu_fwd(zs)
    pooled_ref, pcache = O.maxpool_fwd(as_)
    chk("stem pooled map", val(tr.pooled16), pooled_ref, 5e-4 if X3 else 4e-3)
    g0 = gval(taps[0][2])                                           # the gradient block 0 handed to the stem
    dys, dgs, dbs_ = O.batchnorm_bwd(O.relu_bwd(O.maxpool_bwd(g0, pcache), as_), bcache)
    _, dws, _ = O.conv2d_bwd(dys, ccache)
    chk("stem dw", G[tr.stem.kernel.name], dws, 2e-2)
    chk("stem bn dgamma", G[tr.stem_bn.gamma.name], dgs, 2e-2); chk("stem bn dbeta", G[tr.stem_bn.beta.name], dbs_, 1e-2)
    assert np.abs(G[tr.stem.bias.name]).max() == 0.0               # feeds a BatchNorm: exactly zero, not computed
    assert tr.x3 == X3
    print("%s trunk %s B=%d T=%d: worst block-local error %.3e (%s)" % (mode, res_type, B, T, max(errs.values()), max(errs, key=errs.get)))
    X3 = False; GS = 0.0
