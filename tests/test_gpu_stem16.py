"""GPU parity of the recompute-form stem of the bf16 trunk (csrc/stem_bf16.hip) against the float64 oracle (oracle/ops_np.py):
Conv2D 7x7/2 'same' -> BatchNormalization -> ReLU -> MaxPooling2D 3x3/2 'same' (resnet.py:173-174 / 191-192), forward and backward.

Input and kernel are rounded to the bf16 grid exactly as the kernels round them, so products are exact and what is left is fp32
accumulation order, the bf16 rounding of the stored pooled map (<= 4e-3 of max) and -- backward -- the bf16 rounding of the
per-pixel gradient in front of the second MFMA and xhat recovered from the bf16 pooled value (<= 1e-2 of max on sums over
all pixels).  Shapes: the bench shape (400 x 80, C0 = 64), res18's 32-channel stem, odd sizes (pool padding in front), a ragged
last tile, a time axis shorter than one tile.
"""
import numpy as np
import pytest
import torch

from oracle import ops_np as O

pytestmark = pytest.mark.gpu
BF, SFX, TOL_ST = torch.bfloat16, "", 4e-3
EPS, MOM = 1e-3, 0.99


@pytest.fixture(scope="module", autouse=True, params=["bf16", "fp16"])
def two_byte_type(request):
    """Both builds of csrc/stem_bf16.hip: bf16 (sar_stem16_*) and fp16 (sar_stem16_*_f16, the same source with IEEE half)."""
    global BF, SFX, TOL_ST
    old = (BF, SFX, TOL_ST)
    BF, SFX, TOL_ST = (torch.float16, "_f16", 5e-4) if request.param == "fp16" else (torch.bfloat16, "", 4e-3)
    yield request.param
    BF, SFX, TOL_ST = old


@pytest.fixture(scope="module")
def L():
    from aesrc2020_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.lib()
    return _lib


def rbf(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32).to(BF).double().numpy()


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")


def rel(got, ref):
    got = got.detach().float().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.all(np.isfinite(got))
    return np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)


def run_case(L, B, T, D, C0, seed, train=True):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, (B, T, D)).astype(np.float32)
    w = (rng.standard_normal((7, 7, 1, C0)) * np.sqrt(2.0 / 49)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(C0)).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, C0).astype(np.float32); beta = (0.3 * rng.standard_normal(C0)).astype(np.float32)
    Ho, pt, _ = O.same_pads(T, 7, 2); Wo, pl, _ = O.same_pads(D, 7, 2)
    Hp, ppt, _ = O.same_pads(Ho, 3, 2); Wp, ppl, _ = O.same_pads(Wo, 3, 2)
    NB = L.bn_bins()
    d = dict(x=dev(x), w=dev(w), b=dev(bias), gamma=dev(gamma), beta=dev(beta), bins=torch.zeros(NB, 2, C0, device="cuda"),
             bbins=torch.zeros(NB, 2, C0, device="cuda"), saved=torch.zeros(4, C0, device="cuda"), mm=torch.zeros(C0, device="cuda"),
             mv=torch.ones(C0, device="cuda"), dg=torch.zeros(C0, device="cuda"), db=torch.zeros(C0, device="cuda"))
    R = B * Ho * Wo

    def ref(stats, write=True):
        r = L.BnRef()
        r.gamma, r.beta, r.saved = d["gamma"].data_ptr(), d["beta"].data_ptr(), d["saved"].data_ptr()
        r.bins = d["bins"].data_ptr() if stats == "fwd" else None
        r.bbins = d["bbins"].data_ptr() if stats == "bwd" else None
        r.moving_mean, r.moving_var, r.dgamma, r.dbeta = d["mm"].data_ptr(), d["mv"].data_ptr(), d["dg"].data_ptr(), d["db"].data_ptr()
        r.inv_n, r.unbias, r.eps, r.momentum, r.write = 1.0 / R, R / max(R - 1.0, 1.0), EPS, MOM, int(write)
        r.ginv = 1.0
        return r

    conv = (B, T, D, Ho, Wo, C0, pt, pl); pool = (Hp, Wp, ppt, ppl)
    # ---- oracle forward
    xq, wq = rbf(x), rbf(w)
    y, ccache = O.conv2d_fwd(xq[..., None], wq, bias, (2, 2), "same")
    assert y.shape == (B, Ho, Wo, C0)
    if train:
        z, bcache = O.batchnorm_fwd(y, gamma, beta, EPS)
        mu, var = bcache[4], bcache[5]
    else:
        mu = 0.05 * rng.standard_normal(C0); var = rng.uniform(0.02, 0.08, C0)
        rstd = 1.0 / np.sqrt(var + EPS)
        z = (y - mu) * rstd * gamma + beta
    a = O.relu_fwd(z)
    pooled_ref, pcache = O.maxpool_fwd(a)
    assert pooled_ref.shape == (B, Hp, Wp, C0)
    # ---- kernels forward
    pooled = torch.zeros(B, Hp, Wp, C0, dtype=BF, device="cuda"); idx = torch.full((B, Hp, Wp, C0), 99, dtype=torch.uint8, device="cuda")
    if train:
        L.call("sar_stem16_stats" + SFX, d["x"], d["w"], d["b"], d["bins"], *conv, 0)
        sums = d["bins"].double().sum(0).cpu().numpy()
        assert rel(sums[0], y.sum((0, 1, 2))) < 1e-4 and rel(sums[1], (y * y).sum((0, 1, 2))) < 1e-4
        L.call("sar_stem16_pool_fwd" + SFX, d["x"], d["w"], d["b"], ref("fwd"), pooled, idx, *conv, *pool)
        sv = d["saved"].double().cpu().numpy()
        assert rel(sv[0], mu) < 1e-4 and rel(sv[1], 1 / np.sqrt(var + EPS)) < 1e-4
        assert rel(d["mm"], (1 - MOM) * mu) < 1e-4 and rel(d["mv"], MOM + (1 - MOM) * var * R / (R - 1)) < 1e-4
    else:
        rstd = 1.0 / np.sqrt(var + EPS)
        d["saved"].copy_(dev(np.stack([mu, rstd, gamma * rstd, beta - mu * gamma * rstd])))
        L.call("sar_stem16_pool_fwd" + SFX, d["x"], d["w"], d["b"], ref("saved", False), pooled, idx, *conv, *pool)
    torch.cuda.synchronize()
    assert rel(pooled, pooled_ref) <= TOL_ST
    # argmax slots: equal wherever the oracle's best and second-best window values are clearly apart
    xp = np.pad(a, ((0, 0), (ppt, 2), (ppl, 2), (0, 0)), constant_values=-np.inf)
    cols = O._im2col(xp, 3, 3, 2, 2, Hp, Wp).reshape(B, Hp, Wp, 9, C0)
    srt = np.sort(cols, 3)
    clear = (srt[:, :, :, -1] - srt[:, :, :, -2]) > 2e-2 * np.abs(pooled_ref).max()
    got_idx = idx.cpu().numpy()
    assert clear.mean() > 0.3
    assert np.array_equal(got_idx[clear], pcache[1][clear])
    assert got_idx.max() <= 8
    if not train:
        return
    # ---- backward: gradient of the pooled map on the bf16 grid
    g = rbf(rng.standard_normal((B, Hp, Wp, C0)))
    da = O.maxpool_bwd(g, pcache)
    dzr = O.relu_bwd(da, a)
    dy, dgamma, dbeta = O.batchnorm_bwd(dzr, bcache)
    _, dw_ref, _ = O.conv2d_bwd(dy, ccache)
    gd = torch.tensor(g, dtype=torch.float32).to(BF).cuda()
    work = torch.empty(L.stem16_wgrad_workspace_bytes(C0) // 4, device="cuda")
    L.call("sar_stem16_bwd_reduce" + SFX, gd, pooled, idx, d["x"], d["w"], d["b"], ref("bwd"), work, *conv, *pool)
    bs = d["bbins"].double().sum(0).cpu().numpy()
    assert rel(bs[0], dbeta) < 2e-3, rel(bs[0], dbeta)
    assert rel(bs[1], dgamma) < 1e-2, rel(bs[1], dgamma)
    dw = torch.ones(7, 7, 1, C0, device="cuda")
    L.call("sar_stem16_wgrad" + SFX, d["x"], d["w"], d["b"], gd, pooled, idx, ref("bwd"), dw, work, *conv, *pool)
    torch.cuda.synchronize()
    e = rel(dw - 1.0, dw_ref)
    assert e < 1e-2, e
    assert rel(d["dg"], dgamma) < 1e-2 and rel(d["db"], dbeta) < 2e-3


@pytest.mark.parametrize("B,T,D,C0", [(4, 400, 80, 64), (3, 100, 80, 32), (2, 131, 37, 64), (2, 9, 80, 64), (2, 203, 80, 32)])
def test_stem16_train(L, B, T, D, C0):
    run_case(L, B, T, D, C0, seed=B * 1000 + T)


def test_stem16_inference_constants(L):
    run_case(L, 2, 123, 80, 64, seed=5, train=False)


def test_stem16_bench_shape(L):
    """The bench geometry (400 x 80, C0 = 64: cfg2) at B=16 -- 1 280 conv tiles, persistent weight-gradient blocks with several
    tiles each; the float64 oracle of B=64 needs ~3 GB, the full-size model tests cover that batch."""
    run_case(L, 16, 400, 80, 64, seed=77)
