"""GPU parity tests: every C-ABI entry point of libsarnet_hip.so against the CPU oracle (oracle/ops_np.py)
on the same seeded inputs.  Tolerance: fp32 kernels vs the float64 oracle, max-abs error relative to the
reference's max-abs value <= 2e-4 (north-star asks 1e-3 on logits / CTC loss)."""
import numpy as np
import pytest
import torch

from oracle import ops_np as O

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.fixture(scope="module")
def L():
    from aesrc2020_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.lib()
    return _lib


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


def zeros(*shape, dtype=torch.float32):
    return torch.zeros(*shape, dtype=dtype, device="cuda")


def check(got, ref, tol=TOL, name="", scale=None):
    """max-abs error relative to `scale` (default: the reference's max-abs value)."""
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.all(np.isfinite(got)), name
    err = np.abs(got - ref).max() / ((np.abs(ref).max() if scale is None else scale) + 1e-6)
    assert err <= tol, "%s: rel-to-max error %.3e > %.1e" % (name, err, tol)


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("a_kc,b_kc", [(1, 0), (1, 1), (0, 0), (0, 1)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (70, 33, 19), (2496, 256, 256), (5, 8, 300)])
def test_gemm_layouts(L, a_kc, b_kc, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)); B = rng.standard_normal((K, N)); bias = rng.standard_normal(N)
    Ad = dev(A if a_kc else A.T); Bd = dev(B.T if b_kc else B); C = zeros(M, N)
    L.call("sar_gemm_f32", Ad, Bd, C, dev(bias), M, N, K, Ad.shape[1], Bd.shape[1], N, a_kc, b_kc, 0, 0, 1)
    check(C, A @ B + bias, name="gemm")
    L.call("sar_gemm_f32", Ad, Bd, C, None, M, N, K, Ad.shape[1], Bd.shape[1], N, a_kc, b_kc, 1, 0, 1)
    check(C, np.tanh(A @ B), name="gemm tanh")
    C0 = rng.standard_normal((M, N)); C = dev(C0)
    L.call("sar_gemm_f32", Ad, Bd, C, None, M, N, K, Ad.shape[1], Bd.shape[1], N, a_kc, b_kc, 0, 1, 1)
    check(C, C0 + A @ B, name="gemm accumulate")
    C = zeros(M, N)
    L.call("sar_gemm_f32", Ad, Bd, C, dev(bias), M, N, K, Ad.shape[1], Bd.shape[1], N, a_kc, b_kc, 0, 0, 4)
    check(C, A @ B + bias, name="gemm splitk")


def test_gemm_tanh_epilogue_is_relative_accurate_and_odd(L):
    """The Dense layers' tanh (DS(..., 'tanh'), model.py:35-42) is evaluated in the GEMM epilogue on v_exp / v_rcp: RELATIVE error <= 2e-6 from
    |z| = 1e-6 to saturation (the first form cancelled near zero -- 1e-3 relative at 1e-4), and tanh(-z) = -tanh(z) bit for bit."""
    z = np.concatenate([np.logspace(-6, 1.2, 2040), [0.0, 0.0625, 0.06249999, 0.5, 1.0, 20.0, 60.0, 88.0]]).astype(np.float32)
    z = np.concatenate([z, -z]).reshape(64, 64)
    A = dev(z); B = dev(np.eye(64)); C = zeros(64, 64)
    L.call("sar_gemm_f32", A, B, C, None, 64, 64, 64, 64, 64, 64, 1, 0, 1, 0, 1)      # exact product with the identity: C = tanh(z)
    got = C.cpu().double().numpy(); ref = np.tanh(z.astype(np.float64))
    nz = ref != 0
    assert np.abs(got[nz] / ref[nz] - 1.0).max() <= 2e-6 and np.all(got[~nz] == 0.0)
    g = got.reshape(-1)
    assert np.array_equal(g[:g.size // 2], -g[g.size // 2:])


@pytest.mark.parametrize("R,I,N,sk", [(250, 64, 96, 1), (2496, 256, 768, 1), (77, 33, 10, 1), (2000, 48, 130, 4), (64, 8192, 256, 1)])
def test_gemm_wgrad_with_fused_bias_gradient(L, R, I, N, sk):
    """Dense backward w.r.t. parameters in one launch: dW += X^T dY and db += column sums of dY (both accumulate)."""
    rng = np.random.default_rng(R + I + N)
    X = rng.standard_normal((R, I)); dY = rng.standard_normal((R, N))
    dW0 = rng.standard_normal((I, N)); db0 = rng.standard_normal(N)
    dW = dev(dW0); db = dev(db0)
    L.call("sar_gemm_wgrad_f32", dev(X), dev(dY), dW, db, I, N, R, I, N, N, sk)
    check(dW, dW0 + X.T @ dY, name="dW"); check(db, db0 + dY.sum(0), name="db (fused colsum)")
    dW2 = dev(dW0)
    L.call("sar_gemm_wgrad_f32", dev(X), dev(dY), dW2, None, I, N, R, I, N, N, sk)
    check(dW2, dW0 + X.T @ dY, name="dW without bias")


@pytest.mark.parametrize("a_kc,b_kc", [(1, 0), (1, 1), (0, 0), (0, 1)])
@pytest.mark.parametrize("M,N,K,sk", [(64, 64, 64, 1), (2496, 768, 256, 1), (256, 768, 2496, 3), (128, 192, 8192, 1),
                                      (2496, 1000, 256, 1), (256, 1000, 2496, 3), (2496, 256, 1000, 1), (128, 100, 68, 1),    # N / K tails (bpe-1k layer)
                                      (6048, 256, 512, 1), (104, 256, 256, 1), (6048, 1000, 512, 1)])   # M tails (cfg5's 6 048 rows)
def test_gemm_bf16_math(L, a_kc, b_kc, M, N, K, sk):
    """sar_set_gemm_math(1): operands rounded to bf16 in the staging, bf16 MFMA, fp32 accumulate -- against float64 on the
    rounded operands (tight) for all four layouts, bias + tanh, accumulate, split-K and the fused column sum."""
    rng = np.random.default_rng(M + N + K + 2 * a_kc + b_kc)
    bf = lambda a: torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).double().numpy()
    A = rng.standard_normal((M, K)); B = rng.standard_normal((K, N)) / np.sqrt(K); bias = rng.standard_normal(N)
    Ad = dev(A if a_kc else A.T.copy()); Bd = dev(B.T.copy() if b_kc else B)
    lda = K if a_kc else M; ldb = K if b_kc else N
    ref = bf(A) @ bf(B)
    old = L.get_gemm_math()
    L.set_gemm_math("bf16")
    try:
        C = zeros(M, N)
        L.call("sar_gemm_f32", Ad, Bd, C, dev(bias), M, N, K, lda, ldb, N, a_kc, b_kc, 1, 0, 1)
        check(C, np.tanh(ref + bias), 2e-5, "bf16 gemm + bias + tanh")
        C0 = rng.standard_normal((M, N)); C = dev(C0)
        L.call("sar_gemm_f32", Ad, Bd, C, None, M, N, K, lda, ldb, N, a_kc, b_kc, 0, 1, sk)
        check(C, C0 + ref, 2e-5, "bf16 gemm accumulate (split-K %d)" % sk)
        if not a_kc and not b_kc:                              # the Dense weight-gradient form with its fused bias gradient
            dW = zeros(M, N); db = zeros(N)
            L.call("sar_gemm_wgrad_f32", Ad, Bd, dW, db, M, N, K, M, N, N, sk)
            check(dW, ref, 2e-5, "bf16 wgrad gemm"); check(db, B.sum(0), 2e-5, "fused column sum (fp32 values)")
    finally:
        L.set_gemm_math(old)
    assert L.get_gemm_math() == old


def test_gemm_wgrad_group(L):
    """sar_gemm_wgrad_group_plan / _f32: weight gradients of different shapes in ONE launch (the deferred Dense / GRU kernel gradients
    of a backward pass: cfg2 / cfg3 shapes incl. the bpe-1k tail, a K = 64 problem, split-K) against float64 on the bf16-rounded
    operands; both results accumulate; a problem without a bias gradient; the plan refuses the fp32 mode and ineligible shapes."""
    rng = np.random.default_rng(17)
    bf = lambda a: torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).double().numpy()
    shapes = [(2496, 768, 256, 1, True), (2496, 256, 768, 3, True), (2496, 256, 768, 3, True), (64, 8192, 256, 1, True), (2496, 512, 256, 2, False),
              (2496, 256, 1000, 2, True), (312, 128, 100, 1, True), (6048, 256, 768, 3, True)]   # (R, in, out, splitk, bias gradient)
    old = L.get_gemm_math()
    L.set_gemm_math("bf16")
    try:
        probs, refs = [], []
        for R, I, N, sk, hb in shapes:
            X = rng.standard_normal((R, I)); dY = rng.standard_normal((R, N)) / np.sqrt(R)
            dW0 = rng.standard_normal((I, N)); db0 = rng.standard_normal(N)
            dW, db = dev(dW0), (dev(db0) if hb else None)
            probs.append((dev(X), dev(dY), dW, db, I, N, R, I, N, N, sk))
            refs.append((dW0 + bf(X).T @ bf(dY), db0 + dY.sum(0)))
        grp = L.GemmWgradGroup(probs)
        assert grp.n == len(shapes) and grp.blocks == sum((-(-N // 64) * (I // 64) * sk + 7) // 8 * 8 for R, I, N, sk, hb in shapes)   # every problem starts on XCD 0
        grp.launch()
        for q, (rw, rb), sh in zip(probs, refs, shapes):
            check(q[2], rw, 2e-5, "grouped dW %r" % (sh,))
            if q[3] is not None:
                check(q[3], rb, 2e-5, "grouped db %r" % (sh,))
        with pytest.raises(ValueError):                                      # a 34-column layer (rows are not 16-byte pieces)
            L.GemmWgradGroup([(zeros(128, 64), zeros(128, 34), zeros(64, 34), None, 64, 34, 128, 64, 34, 34, 1)])
        L.set_gemm_math("fp32")
        with pytest.raises(ValueError):                                      # fp32 GEMM mode: no grouped launch
            L.GemmWgradGroup(probs[:1])
    finally:
        L.set_gemm_math(old)


@pytest.mark.parametrize("math,scale", [("bf16", 0.0), ("bf16x3", 0.0), ("bf16x3", 4096.0), ("bf16", 4096.0)])
@pytest.mark.parametrize("b_kc", [1, 0])
def test_gemm_pair(L, math, scale, b_kc):
    """sar_gemm_pair_f32: C = [C +] A B + A2 B2 in one launch (the BiGRU's input gradient: both directions' products), in the GEMM arithmetic in
    force (scale 0) and in the split-precision mode's BACKWARD arithmetic (scale > 0: single-pass fp16 operands, the gradient operand under
    the scale) -- against float64 products of operands rounded the way the kernel rounds them; small gradient-sized values included."""
    rng = np.random.default_rng(11 + b_kc)
    M, N, K = 2496, 256, 768
    h16 = lambda a, sc=1.0: (torch.tensor(a * sc, dtype=torch.float32).to(torch.float16).double().numpy() / sc)
    bf = lambda a: torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).double().numpy()
    ra, rb, tol = ((lambda a: h16(a, scale)), h16, 2e-5) if scale > 0 else ((bf, bf, 2e-5) if math == "bf16" else ((lambda a: a), (lambda a: a), 3e-5))
    A = [rng.standard_normal((M, K)) * 1e-4 * np.exp(rng.uniform(-6, 0, (M, 1))) for _ in range(2)]      # gradient-sized rows: 2.5e-7 .. 1e-4
    W = [rng.standard_normal((N, K) if b_kc else (K, N)) / 16 for _ in range(2)]
    C0 = rng.standard_normal((M, N)) * 1e-4
    prod = lambda i: ra(A[i]) @ (rb(W[i]).T if b_kc else rb(W[i]))
    old = L.get_gemm_math()
    L.set_gemm_math(math)
    try:
        Ad, Wd = [dev(a) for a in A], [dev(w) for w in W]
        ldb = K if b_kc else N
        C = dev(C0)
        L.call("sar_gemm_pair_f32", Ad[0], Wd[0], Ad[1], Wd[1], C, M, N, K, K, ldb, N, b_kc, 0, scale, 1.0)
        check(C, prod(0) + prod(1), tol, "pair, written")
        C = dev(C0)
        L.call("sar_gemm_pair_f32", Ad[0], Wd[0], Ad[1], Wd[1], C, M, N, K, K, ldb, N, b_kc, 1, scale, 1.0)
        check(C, C0 + prod(0) + prod(1), tol, "pair, accumulated")
        C = dev(C0)
        L.call("sar_gemm_pair_f32", Ad[0], Wd[0], None, None, C, M, N, K, K, ldb, N, b_kc, 0, scale, 1.0)
        check(C, prod(0), tol, "single product")
        with pytest.raises(RuntimeError):                                      # not whole tiles: the caller keeps sar_gemm_f32
            L.call("sar_gemm_pair_f32", Ad[0], Wd[0], None, None, C, M - 4, N, K, K, ldb, N, b_kc, 0, scale, 1.0)
        L.set_gemm_math("fp32")
        with pytest.raises(RuntimeError):
            L.call("sar_gemm_pair_f32", Ad[0], Wd[0], None, None, C, M, N, K, K, ldb, N, b_kc, 0, scale, 1.0)
    finally:
        L.set_gemm_math(old)


def test_gemm_wgrad_group_fp16_backward(L):
    """sar_gemm_wgrad_group_scaled_f32 with gscale > 0 (the split-precision mode's backward arithmetic): dW from fp16 copies of X and of
    gscale * dY, db from the fp32 dY, against float64 on operands rounded that way; tails and split-K as in the grouped launch's own test."""
    rng = np.random.default_rng(13)
    gs = 4096.0
    h16 = lambda a, sc=1.0: (torch.tensor(a * sc, dtype=torch.float32).to(torch.float16).double().numpy() / sc)
    shapes = [(2496, 768, 256, 1, True), (2496, 256, 768, 3, True), (64, 8192, 256, 1, True), (2496, 256, 1000, 2, True), (6048, 256, 768, 3, False)]
    old = L.get_gemm_math()
    L.set_gemm_math("bf16x3")
    try:
        probs, refs = [], []
        for R, I, N, sk, hb in shapes:
            X = rng.standard_normal((R, I)); dY = rng.standard_normal((R, N)) * 1e-4 * np.exp(rng.uniform(-6, 0, (R, 1)))
            dW0 = rng.standard_normal((I, N)) * 1e-3; db0 = rng.standard_normal(N) * 1e-3
            dW, db = dev(dW0), (dev(db0) if hb else None)
            shift = 6 if R < 512 else 0                                      # a per-utterance layer: its own, lower scale (large gradient rows)
            if shift:
                dY = dY * 3e5                                                    # up to ~30: would overflow fp16 under the full scale
            probs.append((dev(X), dev(dY), dW, db, I, N, R, I, N, N, sk, shift))
            refs.append((dW0 + h16(X).T @ h16(dY, gs / 2 ** shift), db0 + dY.sum(0)))
        grp = L.GemmWgradGroup(probs)
        grp.launch(gs)
        for q, (rw, rb), sh in zip(probs, refs, shapes):
            check(q[2], rw, 2e-5, "grouped fp16 dW %r" % (sh,))
            if q[3] is not None:
                check(q[3], rb, 2e-5, "grouped db %r" % (sh,))
    finally:
        L.set_gemm_math(old)


@pytest.mark.parametrize("math", ["fp32", "bf16"])
def test_gemm_batched(L, math):
    """sar_gemm_batched_f32 / sar_gemm_wgrad_batched_f32: up to four equal-shape problems in one launch (the BiGRU's projections
    and kernel gradients) against float64 per problem, both arithmetic modes, split-K included."""
    rng = np.random.default_rng(5)
    bf = (lambda a: torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).double().numpy()) if math == "bf16" else (lambda a: a)
    tol = 2e-5 if math == "bf16" else 2e-4
    R, I, N = 2496, 256, 768
    old = L.get_gemm_math()
    L.set_gemm_math(math)
    try:
        X = rng.standard_normal((R, I)); Ws = [rng.standard_normal((I, N)) / 16 for _ in range(2)]; bs = [rng.standard_normal(N) for _ in range(2)]
        Xd = dev(X); Wd = [dev(w) for w in Ws]; bd = [dev(b) for b in bs]; Cs = [zeros(R, N) for _ in range(2)]
        gb = L.GemmBatch([(Xd, Wd[i], Cs[i], bd[i]) for i in range(2)])
        L.call("sar_gemm_batched_f32", gb, 2, R, N, I, I, N, N, 1, 0, 0, 0, 1)
        for i in range(2):
            check(Cs[i], bf(X) @ bf(Ws[i]) + bs[i], tol, "batched projection %d" % i)
        Xs = [rng.standard_normal((R, I)) for _ in range(4)]; dYs = [rng.standard_normal((R, N)) / 50 for _ in range(4)]
        dW0 = [rng.standard_normal((I, N)) for _ in range(4)]; db0 = [rng.standard_normal(N) for _ in range(4)]
        dW = [dev(a) for a in dW0]; db = [dev(a) for a in db0[:3]] + [None]
        Xsd = [dev(a) for a in Xs]; dYd = [dev(a) for a in dYs]
        gb = L.GemmBatch([(Xsd[i], dYd[i], dW[i], db[i]) for i in range(4)])
        L.call("sar_gemm_wgrad_batched_f32", gb, 4, I, N, R, I, N, N, 3)
        for i in range(4):
            check(dW[i], dW0[i] + bf(Xs[i]).T @ bf(dYs[i]), tol, "batched dW %d" % i)
            if db[i] is not None:
                check(db[i], db0[i] + dYs[i].sum(0), tol, "batched db %d" % i)
    finally:
        L.set_gemm_math(old)


@pytest.mark.parametrize("B,H,D1,D2,A", [(64, 256, 64, 64, 8), (5, 32, 64, 64, 8), (130, 64, 48, 33, 11)])
def test_accent_classifier_fused(L, B, H, D1, D2, A):
    """sar_accent_classifier_f32: Dense(relu) -> Dense(relu) -> Dense(softmax) -> categorical_crossentropy, forward + loss + backward in
    one launch, against the float64 oracle ops (dense / softmax / Keras cross-entropy); all three call modes; several row blocks."""
    rng = np.random.default_rng(B + H)
    x = rng.standard_normal((B, H)); Ws = [rng.standard_normal(s_) / np.sqrt(s_[0]) for s_ in ((H, D1), (D1, D2), (D2, A))]
    bs = [0.1 * rng.standard_normal(n) for n in (D1, D2, A)]
    y = np.eye(A)[rng.integers(0, A, B)]; dl = rng.uniform(0.5, 1.5, B) / B
    h1, c1 = O.dense_fwd(x, Ws[0], bs[0], "relu"); h2, c2 = O.dense_fwd(h1, Ws[1], bs[1], "relu"); p, c3 = O.dense_fwd(h2, Ws[2], bs[2], "softmax")
    loss, dp = O.categorical_crossentropy_fwd_bwd(y, p)
    g, dW3, db3 = O.dense_bwd(dp * dl[:, None], c3); g, dW2, db2 = O.dense_bwd(g, c2); dx, dW1, db1 = O.dense_bwd(g, c1)
    d = [dev(a) for a in (x, Ws[0], bs[0], Ws[1], bs[1], Ws[2], bs[2])]
    outs = lambda: (zeros(B, D1), zeros(B, D2), zeros(B, A), zeros(B))
    g0 = [rng.standard_normal(s_) for s_ in ((H, D1), (D1,), (D1, D2), (D2,), (D2, A), (A,))]
    # predict
    o = outs()
    L.call("sar_accent_classifier_f32", *d, None, None, *o, None, None, None, None, None, None, None, B, H, D1, D2, A)
    check(o[0], h1, name="h1"); check(o[1], h2, name="h2"); check(o[2], p, name="probs")
    # evaluate: + loss
    o = outs()
    L.call("sar_accent_classifier_f32", *d, dev(y), None, *o, None, None, None, None, None, None, None, B, H, D1, D2, A)
    check(o[2], p, name="probs (eval)"); check(o[3], loss, name="loss")
    # train: + gradients (accumulated onto what the buffers held) and dx (written)
    o = outs(); gd = [dev(a) for a in g0]; dxd = dev(rng.standard_normal((B, H)))
    L.call("sar_accent_classifier_f32", *d, dev(y), dev(dl), *o, *gd, dxd, B, H, D1, D2, A)
    check(o[3], loss, name="loss (train)")
    for got, base, ref, nm in zip(gd, g0, (dW1, db1, dW2, db2, dW3, db3), ("dW1", "db1", "dW2", "db2", "dW3", "db3")):
        check(got, base + ref, name=nm)
    check(dxd, dx, name="dx")


@pytest.mark.parametrize("B,H,A,mode", [(64, 256, 8, 2), (64, 256, 8, 0), (5, 32, 4, 1), (130, 192, 8, 2), (3, 256, 8, 1)])
def test_accent_margin_one_launch(L, B, H, A, mode):
    """sar_accent_margin_train_f32 (accent classifier + margin head, forward + losses + backward, ONE launch) against the two calls it
    replaces (each held to the float64 oracle above): every output, dx = the sum of both heads' input gradients, every accumulated
    parameter gradient; three calls in a row rely on the workspace + ticket being left zero; ragged last row block."""
    rng = np.random.default_rng(7 * B + H + mode)
    D1 = D2 = 64
    x = rng.standard_normal((B, H)); Ws = [rng.standard_normal(s_) / np.sqrt(s_[0]) for s_ in ((H, D1), (D1, D2), (D2, A))]
    bs = [0.1 * rng.standard_normal(n) for n in (D1, D2, A)]
    y = np.eye(A)[rng.integers(0, A, B)]; dl = rng.uniform(0.5, 1.5, B) / B; mdl = rng.uniform(0.5, 1.5, B) / B
    Wm = rng.standard_normal((H, A)) / 16; lab = np.eye(A)[rng.integers(0, A, B)]
    s_, m_ = 30.0, (1.35, 0.35, 0.3)[mode]
    d = [dev(a) for a in (x, Ws[0], bs[0], Ws[1], bs[1], Ws[2], bs[2])]
    g0 = [rng.standard_normal(s_) for s_ in ((H, D1), (D1,), (D1, D2), (D2,), (D2, A), (A,))]; dWm0 = rng.standard_normal((H, A))
    yd, dld, mdld, Wmd, labd = dev(y), dev(dl), dev(mdl), dev(Wm), dev(lab)
    # the two calls
    o = (zeros(B, D1), zeros(B, D2), zeros(B, A), zeros(B)); gd = [dev(a) for a in g0]; dxr = zeros(B, H)
    L.call("sar_accent_classifier_f32", *d, yd, dld, *o, *gd, dxr, B, H, D1, D2, A)
    mp, ml, miw, dWmr, ws = zeros(B, A), zeros(B), zeros(A), dev(dWm0), zeros(H, A)
    L.call("sar_margin_head_train_f32", d[0], labd, labd, mdld, Wmd, mp, ml, miw, dxr, dWmr, ws, B, H, A, mode, s_, m_, 1)
    # one launch
    o2 = (zeros(B, D1), zeros(B, D2), zeros(B, A), zeros(B)); gd2 = [dev(a) for a in g0]; dx2 = dev(rng.standard_normal((B, H)))
    mp2, ml2, miw2, dWm2, ws2 = zeros(B, A), zeros(B), zeros(A), dev(dWm0), zeros(H * A + 1)
    for _ in range(3):
        for t, a in zip(gd2 + [dWm2], g0 + [dWm0]):
            t.copy_(dev(a))
        L.call("sar_accent_margin_train_f32", *d, yd, dld, *o2, *gd2, dx2, labd, labd, mdld, Wmd, mp2, ml2, miw2, dWm2, ws2,
               B, H, D1, D2, A, mode, s_, m_)
    ref = lambda t: t.cpu().double().numpy()
    for got, want, nm in zip(o2, o, ("h1", "h2", "probs", "loss")):
        check(got, ref(want), 1e-5, nm)
    for got, want, nm in zip(gd2, gd, ("dW1", "db1", "dW2", "db2", "dW3", "db3")):
        check(got, ref(want), 1e-5, nm)
    check(mp2, ref(mp), 1e-5, "margin probs"); check(ml2, ref(ml), 1e-5, "margin loss"); check(miw2, ref(miw), 1e-5, "inv_w")
    check(dx2, ref(dxr), 1e-4, "dx"); check(dWm2, ref(dWmr), 1e-4, "dWm")
    assert float(ws2.abs().max()) == 0.0
    # 200 more calls back to back: the batch's sum crosses workgroups (and XCDs) through device-scope atomics + a ticket inside the launch --
    # every call must collect all of it (dWm grows by exactly one increment per call) and leave the workspace and the ticket zero
    inc = ref(dWmr) - dWm0
    dWm2.zero_()
    for _ in range(200):
        L.call("sar_accent_margin_train_f32", *d, yd, dld, *o2, *gd2, dx2, labd, labd, mdld, Wmd, mp2, ml2, miw2, dWm2, ws2,
               B, H, D1, D2, A, mode, s_, m_)
    check(dWm2, 200 * inc, 1e-4, "dWm after 200 calls"); check(dx2, ref(dxr), 1e-4, "dx after 200 calls")
    assert float(ws2.abs().max()) == 0.0
    # sizes outside the register-cached margin form are refused (the caller keeps the two calls)
    with pytest.raises(Exception):
        L.call("sar_accent_margin_train_f32", *d, yd, dld, *o2, *gd2, dx2, labd, labd, mdld, Wmd, mp2, ml2, miw2, dWm2, ws2,
               B, H, D1, D2, 6, mode, s_, m_)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_margin_head_train_fused(L, mode):
    """sar_margin_head_train_f32 (forward + cross-entropy + backward in two launches) against the per-op entry points it replaces
    (those are held to the float64 oracle by test_margin_heads): probabilities, loss, dx (accumulated), dW; workspace left zero."""
    rng = np.random.default_rng(40 + mode)
    B, D, A = 64, 256, 8
    x = rng.standard_normal((B, D)); W = rng.standard_normal((D, A)) / 16; lab = np.eye(A)[rng.integers(0, A, B)]
    dl = rng.uniform(0.5, 1.5, B) / B; dx0 = rng.standard_normal((B, D)); dW0 = rng.standard_normal((D, A))
    s_, m_ = 30.0, (1.35, 0.35, 0.3)[mode]
    xd, Wd, ld, dld = dev(x), dev(W), dev(lab), dev(dl)
    # reference: the separate launches
    probs, cosv, invx, invw, loss, dp = zeros(B, A), zeros(B, A), zeros(B), zeros(A), zeros(B), zeros(B, A)
    L.call("sar_margin_head_fwd_f32", xd, ld, Wd, probs, cosv, invx, invw, B, D, A, mode, s_, m_)
    L.call("sar_cce_fwd_f32", ld, probs, loss, B, A); L.call("sar_cce_bwd_f32", ld, probs, dld, dp, B, A)
    dx_ref, dW_ref, ws = dev(dx0), dev(dW0), zeros(D, A)
    L.call("sar_margin_head_bwd_f32", dp, probs, cosv, xd, ld, Wd, invx, invw, dx_ref, dW_ref, ws, B, D, A, mode, s_, m_, 1)
    # fused
    probs2, loss2, invw2, dx2, dW2, ws2 = zeros(B, A), zeros(B), zeros(A), dev(dx0), dev(dW0), zeros(D, A)
    for _ in range(2):                                        # twice: the second call relies on the workspace being left zero
        dx2.copy_(dev(dx0)); dW2.copy_(dev(dW0))
        L.call("sar_margin_head_train_f32", xd, ld, ld, dld, Wd, probs2, loss2, invw2, dx2, dW2, ws2, B, D, A, mode, s_, m_, 1)
    check(probs2, probs.cpu().double().numpy(), 1e-5, "probs"); check(loss2, loss.cpu().double().numpy(), 1e-5, "loss")
    check(dx2, dx_ref.cpu().double().numpy(), 1e-4, "dx"); check(dW2, dW_ref.cpu().double().numpy(), 1e-4, "dW")
    assert float(ws2.abs().max()) == 0.0


def test_gemm_transpose_detecting(L):
    # asymmetric B with A = I catches a swapped C write (guide rule 16)
    n = 64
    B = np.arange(n * n, dtype=np.float64).reshape(n, n) / n
    C = zeros(n, n)
    L.call("sar_gemm_f32", dev(np.eye(n)), dev(B), C, None, n, n, n, n, n, n, 1, 0, 0, 0, 1)
    check(C, B, 1e-6)


# ------------------------------------------------------------------ conv
CONV_CASES = [  # B,H,W,Cin,Cout,k,stride,pad,prologue,residual
    (2, 10, 8, 16, 16, 3, 1, "same", True, True), (2, 10, 8, 32, 64, 3, 2, "same", True, False),
    (2, 13, 5, 16, 32, 3, 2, "same", False, False), (3, 25, 5, 64, 128, 1, 2, "valid", False, False),
    (2, 12, 6, 64, 32, 1, 1, "valid", False, False), (1, 20, 20, 32, 32, 3, 1, "same", True, False),
    # narrow layers on the split-precision path: Cout (fwd) / Cin (dgrad) below the 32-wide tile
    (2, 25, 20, 64, 16, 3, 1, "same", True, True), (2, 25, 20, 32, 16, 3, 1, "same", True, False),
    (2, 26, 20, 16, 32, 3, 2, "same", True, False), (2, 25, 20, 16, 64, 3, 1, "same", False, False),
    (2, 13, 3, 128, 128, 3, 1, "same", True, True), (2, 10, 6, 64, 128, 3, 2, "same", True, False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(L, case):
    B, H, W, Cin, Cout, k, s, pad, pro, res = case
    rng = np.random.default_rng(sum(case[:7]))
    x = rng.standard_normal((B, H, W, Cin)); w = rng.standard_normal((k, k, Cin, Cout)) * 0.1; b = rng.standard_normal(Cout)
    scale = rng.uniform(0.5, 1.5, Cin); shift = rng.standard_normal(Cin) * 0.3
    act = np.maximum(x * scale + shift, 0) if pro else x
    y_ref, cache = O.conv2d_fwd(act, w, b, (s, s), pad)
    _, Ho, Wo, _ = y_ref.shape
    if pad == "same":
        pt = O.same_pads(H, k, s)[1]; pl = O.same_pads(W, k, s)[1]
    else:
        pt = pl = 0
    r = rng.standard_normal(y_ref.shape) if res else None
    xd, wd = dev(x), dev(w)
    scd, shd = (dev(scale), dev(shift)) if pro else (None, None)
    y = zeros(B, Ho, Wo, Cout)
    L.call("sar_conv2d_fwd_f32", xd, wd, dev(b), scd, shd, dev(r) if res else None, y, B, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl)
    check(y, y_ref + (r if res else 0), name="conv fwd")
    dy = rng.standard_normal(y_ref.shape)
    dact_ref, dw_ref, db_ref = O.conv2d_bwd(dy, cache)
    dyd = dev(dy)
    dx = zeros(B, H, W, Cin)
    L.call("sar_conv2d_dgrad_f32", dyd, wd, dx, B, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, 0)
    check(dx, dact_ref, name="conv dgrad")
    L.call("sar_conv2d_dgrad_f32", dyd, wd, dx, B, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, 1)
    check(dx, 2 * dact_ref, name="conv dgrad accumulate")
    dw = zeros(k, k, Cin, Cout); db = zeros(Cout)
    L.call("sar_conv2d_wgrad_f32", xd, dyd, scd, shd, dw, db, B, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl)
    check(dw, dw_ref, name="conv wgrad"); check(db, db_ref, name="conv dbias (fused)")
    dw2 = zeros(k, k, Cin, Cout)
    L.call("sar_conv2d_wgrad_f32", xd, dyd, scd, shd, dw2, None, B, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl)
    check(dw2, dw_ref, name="conv wgrad (no bias)")
    db2 = zeros(Cout)
    L.call("sar_colsum_f32", dyd, db2, B * Ho * Wo, Cout)
    check(db2, db_ref, name="colsum")


# Every distinct conv of the bench trunk (res34, F=32, T=400: pooled map 100x20x64) at the bench batch B=64, i.e. the launch
# configurations bench.py actually runs in the fp32-storage modes: multi-slice weight gradients on the XCD-grouped grid,
# split-K + pre-split weight planes in stages 2-3 (resnet.py:96-125).  Held per tensor to max-abs <= 2e-4 (fp32 MFMA) /
# <= 1e-4 (bf16x3) of the tensor's max -- no flat L2.
BENCH_CONVS = [  # H, W, Cin, Cout, k, stride, pad, prologue+residual
    (100, 20, 64, 32, 3, 1, "same", False), (100, 20, 64, 32, 1, 1, "valid", False), (100, 20, 32, 32, 3, 1, "same", True),
    (100, 20, 32, 64, 3, 2, "same", True), (100, 20, 32, 64, 1, 2, "valid", False), (50, 10, 64, 64, 3, 1, "same", True),
    (50, 10, 64, 128, 3, 2, "same", True), (50, 10, 64, 128, 1, 2, "valid", False), (25, 5, 128, 128, 3, 1, "same", True),
    (25, 5, 128, 256, 3, 2, "same", True), (25, 5, 128, 256, 1, 2, "valid", False), (13, 3, 256, 256, 3, 1, "same", True),
]


@pytest.mark.parametrize("math", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", BENCH_CONVS, ids=["%dx%dx%d-%d_k%d_s%d" % c[:6] for c in BENCH_CONVS])
def test_conv_bench_shapes_at_batch_64(L, case, math):
    import ctypes
    H, W, Cin, Cout, k, s, pad, full = case
    B = 64
    tol = 2e-4 if math == "fp32" else 1e-4
    rng = np.random.default_rng(sum(case[:6]))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32).astype(np.float64)
    w = (rng.standard_normal((k, k, Cin, Cout)) * (1.0 / np.sqrt(k * k * Cin))).astype(np.float32).astype(np.float64)
    b = rng.standard_normal(Cout).astype(np.float32).astype(np.float64)
    scale = rng.uniform(0.5, 1.5, Cin).astype(np.float32).astype(np.float64); shift = (rng.standard_normal(Cin) * 0.3).astype(np.float32).astype(np.float64)
    act = np.maximum(x * scale + shift, 0) if full else x
    y_ref, cache = O.conv2d_fwd(act, w, b, (s, s), pad)
    _, Ho, Wo, _ = y_ref.shape
    pt, pl = (O.same_pads(H, k, s)[1], O.same_pads(W, k, s)[1]) if pad == "same" else (0, 0)
    r = rng.standard_normal(y_ref.shape).astype(np.float32).astype(np.float64) if full else None
    dy = rng.standard_normal(y_ref.shape).astype(np.float32).astype(np.float64)
    dact_ref, dw_ref, db_ref = O.conv2d_bwd(dy, cache)
    del cache
    old = L.get_conv_math()
    L.set_conv_math(math)
    try:
        xd, wd, dyd = dev(x), dev(w), dev(dy)
        scd, shd = (dev(scale), dev(shift)) if full else (None, None)
        # the train step's form for the wide layers: kernels pre-split into bf16 planes (forward / input gradient)
        planes = math == "bf16x3" and Cin % 32 == 0 and Cout % 64 == 0
        wt = wdp = (None, None)
        if planes:
            K = k * k * Cin
            wt = tuple(torch.empty(Cout, K, dtype=torch.bfloat16, device="cuda") for _ in range(2))
            wdp = tuple(torch.empty(K, Cout, dtype=torch.bfloat16, device="cuda") for _ in range(2))
            tab = torch.tensor([[wd.data_ptr(), wt[0].data_ptr(), wt[1].data_ptr(), wdp[0].data_ptr(), wdp[1].data_ptr(), K, Cout, 0]], dtype=torch.int64, device="cuda")
            L.call("sar_conv_weight_split_bf16", tab, 1, (K // 32) * (Cout // 64))
        y = zeros(B, Ho, Wo, Cout)
        L.call("sar_conv2d_fwd_ex_f32", xd, wd, dev(b), scd, shd, dev(r) if full else None, y, B, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl,
               wt[0], wt[1], None, None, None, None, None, None, None, None, None, 1e-3, 0.99, 1)
        check(y, y_ref + (r if full else 0), tol, "conv fwd")
        dx = zeros(B, H, W, Cin)
        L.call("sar_conv2d_dgrad_ex_f32", dyd, wd, dx, B, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, 0, wdp[0], wdp[1])
        check(dx, dact_ref, tol, "conv dgrad")
        dw = zeros(k, k, Cin, Cout); db = zeros(Cout)
        L.call("sar_conv2d_wgrad_f32", xd, dyd, scd, shd, dw, db, B, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl)
        check(dw, dw_ref, tol, "conv wgrad"); check(db, db_ref, tol, "conv dbias")
        info = (ctypes.c_int * 4)()
        assert L.lib().fn["sar_dbg_last_wgrad_launch"](info) == 0
        slices, xcd, x3, rows = list(info)
        assert x3 == (1 if math == "bf16x3" else 0)
        if math == "bf16x3" and k == 3 and Cout >= 64 and B * Ho * Wo >= 8000:
            # the dominant launch class of the fp32-storage profile: many pixel slices, XCD-grouped 1-D grid, atomic flush
            assert slices > 8 and xcd == 1, (slices, xcd, rows)
    finally:
        L.set_conv_math(old)


@pytest.mark.parametrize("case", [  # B,H,W,Cin,Cout,k,stride -- kernels pre-split into bf16 planes (two convs in one table)
    (2, 10, 8, 64, 64, 3, 1), (2, 13, 5, 64, 128, 3, 2), (3, 25, 5, 64, 128, 1, 2), (64, 50, 12, 64, 64, 3, 1), (2, 7, 3, 128, 256, 3, 1)])
def test_conv_with_weight_planes(L, case):
    """sar_conv_weight_split_bf16 + the *_ex conv entry points: planes reproduce the kernel to 2^-16, wt is the transpose of wd,
    and forward / dgrad with planes match the oracle like the converting kernels do."""
    B, H, W, Cin, Cout, k, s = case
    rng = np.random.default_rng(sum(case) + 3)
    x = rng.standard_normal((B, H, W, Cin)); w = rng.standard_normal((k, k, Cin, Cout)) * 0.1; b = rng.standard_normal(Cout)
    w_other = rng.standard_normal((1, 1, 32, 64))                 # a second row in the table
    sc = rng.uniform(0.5, 1.5, Cin); sh = rng.standard_normal(Cin) * 0.3
    pad = "same" if k == 3 else "valid"
    y_ref, cache = O.conv2d_fwd(np.maximum(x * sc + sh, 0), w, b, (s, s), pad)
    _, Ho, Wo, _ = y_ref.shape
    pt, pl = (O.same_pads(H, k, s)[1], O.same_pads(W, k, s)[1]) if pad == "same" else (0, 0)
    wd_, wo_ = dev(w), dev(w_other)
    rows, planes, blk = [], [], 0
    for t in (wo_, wd_):
        K_, C_ = t.numel() // t.shape[-1], t.shape[-1]
        pl_ = [torch.empty(C_, K_, dtype=torch.bfloat16, device="cuda") for _ in range(2)] + \
              [torch.empty(K_, C_, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
        rows.append([t.data_ptr()] + [q.data_ptr() for q in pl_] + [K_, C_, blk]); planes.append(pl_); blk += (K_ // 32) * (C_ // 64)
    L.call("sar_conv_weight_split_bf16", torch.tensor(rows, dtype=torch.int64, device="cuda"), 2, blk)
    for t, (wt_h, wt_l, wd_h, wd_l) in zip((wo_, wd_), planes):
        w2 = t.reshape(-1, t.shape[-1])
        assert float(((wd_h.float() + wd_l.float()) - w2).abs().max()) <= 2.0 ** -16 * float(w2.abs().max())
        assert torch.equal(wd_h, w2.to(torch.bfloat16)) and torch.equal(wt_h, wd_h.t().contiguous()) and torch.equal(wt_l, wd_l.t().contiguous())
    wt_h, wt_l, wd_h, wd_l = planes[1]
    y = zeros(B, Ho, Wo, Cout)
    L.call("sar_conv2d_fwd_ex_f32", dev(x), wd_, dev(b), dev(sc), dev(sh), None, y, B, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, wt_h, wt_l,
           None, None, None, None, None, None, None, None, None, 1e-3, 0.99, 1)
    check(y, y_ref, name="conv fwd (weight planes)")
    dy = rng.standard_normal(y_ref.shape)
    dact_ref, _, _ = O.conv2d_bwd(dy, cache)
    dx = zeros(B, H, W, Cin)
    L.call("sar_conv2d_dgrad_ex_f32", dev(dy), wd_, dx, B, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, 0, wd_h, wd_l)
    check(dx, dact_ref, name="conv dgrad (weight planes)")
    L.call("sar_conv2d_dgrad_ex_f32", dev(dy), wd_, dx, B, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, 1, wd_h, wd_l)
    check(dx, 2 * dact_ref, name="conv dgrad (weight planes) accumulate")


@pytest.mark.parametrize("case", [  # B,H,W,Cin,Cout,k,stride,residual -- with / without split-K, narrow / wide, 1x1
    (4, 25, 20, 64, 32, 3, 1, True), (16, 25, 10, 64, 64, 3, 1, True), (3, 13, 3, 128, 128, 3, 1, False),
    (2, 9, 5, 32, 48, 3, 2, False), (64, 50, 10, 64, 64, 3, 1, True), (5, 25, 5, 64, 128, 1, 2, False)])
def test_conv_with_fused_batchnorm_statistics(L, case):
    """sar_conv2d_fwd_bnstats_f32 == sar_conv2d_fwd_f32 followed by sar_bn_stats_f32 on its output (oracle: batchnorm of the
    float64 conv output): mean / rstd / folded scale+shift and the moving averages of the consuming BatchNorm."""
    B, H, W, Cin, Cout, k, s, res = case
    rng = np.random.default_rng(sum(case[:7]) + 7)
    x = rng.standard_normal((B, H, W, Cin)); w = rng.standard_normal((k, k, Cin, Cout)) * 0.1; b = rng.standard_normal(Cout)
    sc = rng.uniform(0.5, 1.5, Cin); sh = rng.standard_normal(Cin) * 0.3
    gamma = rng.uniform(0.5, 1.5, Cout); beta = rng.standard_normal(Cout)
    pad = "same" if k == 3 else "valid"
    y_ref, _ = O.conv2d_fwd(np.maximum(x * sc + sh, 0), w, b, (s, s), pad)
    _, Ho, Wo, _ = y_ref.shape
    pt, pl = (O.same_pads(H, k, s)[1], O.same_pads(W, k, s)[1]) if pad == "same" else (0, 0)
    r = rng.standard_normal(y_ref.shape) if res else None
    if res:
        y_ref = y_ref + r
    R = B * Ho * Wo
    mu = y_ref.reshape(R, Cout).mean(0); var = y_ref.reshape(R, Cout).var(0)
    y = zeros(B, Ho, Wo, Cout)
    mean, rstd, scale, shift = (zeros(Cout) for _ in range(4))
    mm = zeros(Cout); mv = dev(np.ones(Cout))
    sums = zeros(L.bn_workspace_doubles(Cout), dtype=torch.float64)
    L.call("sar_conv2d_fwd_bnstats_f32", dev(x), dev(w), dev(b), dev(sc), dev(sh), dev(r) if res else None, y, B, H, W, Cin, Ho, Wo,
           Cout, k, k, s, pt, pl, dev(gamma), dev(beta), sums, mean, rstd, scale, shift, mm, mv, 1e-3, 0.99, 1)
    check(y, y_ref, name="conv out")
    check(mean, mu, name="bn mean", scale=np.abs(y_ref).max()); check(rstd, 1 / np.sqrt(var + 1e-3), name="bn rstd")
    check(scale, gamma / np.sqrt(var + 1e-3), name="bn scale"); check(shift, beta - mu * gamma / np.sqrt(var + 1e-3), name="bn shift", scale=np.abs(y_ref).max())
    check(mm, 0.01 * mu, name="moving mean", scale=np.abs(y_ref).max() * 0.01)
    check(mv, 0.99 + 0.01 * var * R / (R - 1), name="moving var")


@pytest.mark.parametrize("B,T,D,C0", [(2, 100, 80, 64), (2, 37, 21, 8), (1, 64, 80, 32)])
def test_stem_and_fused_maxpool(L, B, T, D, C0):
    rng = np.random.default_rng(T)
    x = rng.random((B, T, D, 1)); w = rng.standard_normal((7, 7, 1, C0)) * 0.2; b = rng.standard_normal(C0) * 0.1
    y_ref, cache = O.conv2d_fwd(x, w, b, (2, 2), "same")
    Ho, pt, _ = O.same_pads(T, 7, 2); Wo, pl, _ = O.same_pads(D, 7, 2)
    xd = dev(x); y = zeros(B, Ho, Wo, C0)
    L.call("sar_stem_conv7x7_fwd_f32", xd, dev(w), dev(b), y, B, T, D, Ho, Wo, C0, pt, pl)
    check(y, y_ref, name="stem fwd")
    # BN(train) + ReLU + MaxPool fused
    gamma = rng.uniform(0.5, 1.5, C0); beta = rng.standard_normal(C0) * 0.2
    bn_ref, bnc = O.batchnorm_fwd(y_ref, gamma, beta)
    a_ref = O.relu_fwd(bn_ref)
    p_ref, pc = O.maxpool_fwd(a_ref)
    Hp, ppt, _ = O.same_pads(Ho, 3, 2); Wp, ppl, _ = O.same_pads(Wo, 3, 2)
    sums = zeros(L.bn_workspace_doubles(C0), dtype=torch.float64)
    mean, rstd, scale, shift = zeros(C0), zeros(C0), zeros(C0), zeros(C0)
    mm, mv = zeros(C0), torch.ones(C0, device="cuda", dtype=torch.float32)
    yd = dev(y_ref)
    L.call("sar_bn_stats_f32", yd, dev(gamma), dev(beta), sums, mean, rstd, scale, shift, mm, mv, B * Ho * Wo, C0, 1e-3, 0.99, 1)
    check(mean, bnc[4], name="bn mean"); check(rstd, bnc[1], name="bn rstd")
    # the same statistics out of the stem conv's epilogue (twice: the partial rows are cleared by every call)
    for rep in range(2):
        y2, sums_f = zeros(B, Ho, Wo, C0), torch.full((L.bn_workspace_doubles(C0),), 7.0, device="cuda", dtype=torch.float64)
        m2, r2, sc2, sh2 = zeros(C0), zeros(C0), zeros(C0), zeros(C0)
        mm2, mv2 = zeros(C0), torch.ones(C0, device="cuda", dtype=torch.float32)
        L.call("sar_stem_conv7x7_fwd_bnstats_f32", xd, dev(w), dev(b), y2, B, T, D, Ho, Wo, C0, pt, pl, dev(gamma), dev(beta), sums_f,
               m2, r2, sc2, sh2, mm2, mv2, 1e-3, 0.99, 1)
        check(y2, y_ref, name="stem fwd (bnstats)"); check(m2, bnc[4], name="stem bn mean (epilogue)"); check(r2, bnc[1], name="stem bn rstd (epilogue)")
        check(sc2, scale.cpu().numpy(), name="stem bn scale (epilogue)"); check(mv2, mv.cpu().numpy(), name="stem moving var (epilogue)")
    pooled = zeros(B, Hp, Wp, C0); idx = zeros(B, Hp, Wp, C0, dtype=torch.uint8)
    L.call("sar_bn_relu_maxpool_fwd_f32", yd, scale, shift, pooled, idx, B, Ho, Wo, C0, Hp, Wp, ppt, ppl)
    check(pooled, p_ref, name="bn+relu+maxpool fwd")
    dpool = rng.standard_normal(p_ref.shape)
    da = O.maxpool_bwd(dpool, pc)
    dbn = O.relu_bwd(da, a_ref)
    dy_ref, dg_ref, dbeta_ref = O.batchnorm_bwd(dbn, bnc)
    dyd = zeros(B, Ho, Wo, C0)
    L.call("sar_maxpool_bn_relu_bwd_f32", dev(dpool), pooled, idx, yd, mean, rstd, scale, shift, sums, dyd, B, Ho, Wo, C0, Hp, Wp, ppt, ppl)
    check(dyd, dy_ref, name="maxpool+bn+relu bwd")
    dg, dbt = zeros(C0), zeros(C0)
    L.call("sar_bn_param_grads_f32", sums, dg, dbt, C0)
    check(dg, dg_ref, name="stem dgamma"); check(dbt, dbeta_ref, name="stem dbeta")
    # stem wgrad
    _, dw_ref, db_ref = O.conv2d_bwd(dy_ref, cache)
    dw, db = zeros(7, 7, 1, C0), zeros(C0)
    L.call("sar_stem_conv7x7_wgrad_f32", xd, dev(dy_ref), dw, db, B, T, D, Ho, Wo, C0, pt, pl)
    check(dw, dw_ref, name="stem wgrad")
    # the BN-backward gradient sums to exactly zero per channel, so scale by the summed magnitudes instead
    check(db, db_ref, name="stem dbias", scale=np.abs(dy_ref).sum((0, 1, 2)).max())
    # fused form: sums only (dx = NULL), then the weight gradient evaluates the pooling/BatchNorm backward per tile
    sums2 = zeros(L.bn_workspace_doubles(C0), dtype=torch.float64)
    L.call("sar_maxpool_bn_relu_bwd_f32", dev(dpool), pooled, idx, yd, mean, rstd, scale, shift, sums2, None, B, Ho, Wo, C0, Hp, Wp, ppt, ppl)
    dg2, dbt2 = zeros(C0), zeros(C0)
    L.call("sar_bn_param_grads_f32", sums2, dg2, dbt2, C0)
    check(dg2, dg_ref, name="stem dgamma (sums only)"); check(dbt2, dbeta_ref, name="stem dbeta (sums only)")
    dw2, db2 = zeros(7, 7, 1, C0), zeros(C0)
    L.call("sar_stem_conv7x7_wgrad_poolbwd_f32", xd, yd, dev(dpool), pooled, idx, mean, rstd, scale, sums2, dw2, db2, B, T, D, Ho, Wo, C0,
           pt, pl, Hp, Wp, ppt, ppl)
    check(dw2, dw_ref, name="stem wgrad (fused pool/BN backward)")
    check(db2, db_ref, name="stem dbias (fused)", scale=np.abs(dy_ref).sum((0, 1, 2)).max())


# ------------------------------------------------------------------ normalisation
@pytest.mark.parametrize("R,C,relu", [(200, 32, 1), (999, 64, 1), (77, 256, 0), (64, 8, 1)])
def test_batchnorm_4d(L, R, C, relu):
    rng = np.random.default_rng(R + C)
    x = rng.standard_normal((R, C)) * 2 + 0.5; gamma = rng.uniform(0.5, 1.5, C); beta = rng.standard_normal(C) * 0.3
    y_ref, cache = O.batchnorm_fwd(x, gamma, beta)
    a_ref = O.relu_fwd(y_ref) if relu else y_ref
    xd = dev(x); sums = zeros(L.bn_workspace_doubles(C), dtype=torch.float64)
    mean, rstd, scale, shift = zeros(C), zeros(C), zeros(C), zeros(C)
    L.call("sar_bn_stats_f32", xd, dev(gamma), dev(beta), sums, mean, rstd, scale, shift, None, None, R, C, 1e-3, 0.99, 1)
    y = zeros(R, C)
    L.call("sar_bn_apply_f32", xd, scale, shift, y, R, C, relu)
    check(y, a_ref, name="bn apply")
    dact = rng.standard_normal((R, C)); add = rng.standard_normal((R, C))
    dz = O.relu_bwd(dact, a_ref) if relu else dact
    dx_ref, dg_ref, db_ref = O.batchnorm_bwd(dz, cache)
    dx, dg, db = zeros(R, C), zeros(C), zeros(C)
    L.call("sar_bn_bwd_f32", dev(dact), xd, scale, shift, mean, rstd, sums, dev(add), dx, dg, db, R, C, relu)
    check(dx, dx_ref + add, name="bn bwd dx"); check(dg, dg_ref, name="bn dgamma"); check(db, db_ref, name="bn dbeta")


@pytest.mark.parametrize("B,C", [(64, 2048), (5, 130), (64, 256), (128, 64), (200, 96)])    # (the last: more rows than a thread keeps in registers)
def test_batchnorm_2d(L, B, C):
    rng = np.random.default_rng(C)
    x = rng.standard_normal((B, C)); gamma = rng.uniform(0.5, 1.5, C); beta = rng.standard_normal(C)
    y_ref, cache = O.batchnorm_fwd(x, gamma, beta)
    xd, gd = dev(x), dev(gamma); y, mean, rstd = zeros(B, C), zeros(C), zeros(C)
    L.call("sar_bn1d_fwd_f32", xd, gd, dev(beta), y, mean, rstd, None, None, B, C, 1e-3, 0.99)
    check(y, y_ref, name="bn1d fwd")
    # moving statistics (model.py:286,289 use): Keras 2.2.x feeds the moving average var * n / (n - (1 + eps))
    mm, mv = zeros(C), dev(np.ones(C)); y2 = zeros(B, C)
    L.call("sar_bn1d_fwd_f32", xd, gd, dev(beta), y2, mean, rstd, mm, mv, B, C, 1e-3, 0.99)
    check(y2, y_ref, name="bn1d fwd (with moving statistics)")
    check(mm, 0.01 * x.mean(0), name="bn1d moving mean"); check(mv, 0.99 + 0.01 * x.var(0) * B / (B - 1.001), name="bn1d moving variance")
    dy = rng.standard_normal((B, C)); dx_ref, dg_ref, db_ref = O.batchnorm_bwd(dy, cache)
    dx, dg, db = zeros(B, C), zeros(C), zeros(C)
    L.call("sar_bn1d_bwd_f32", dev(dy), xd, gd, mean, rstd, dx, dg, db, B, C)
    check(dx, dx_ref, name="bn1d dx"); check(dg, dg_ref, name="bn1d dgamma"); check(db, db_ref, name="bn1d dbeta")


@pytest.mark.parametrize("R,C", [(2496, 256), (2496, 512), (100, 512), (4099, 256), (7, 16), (33, 2048)])
def test_layernorm(L, R, C):
    rng = np.random.default_rng(R)
    x = rng.standard_normal((R, C)); gamma = rng.uniform(0.5, 1.5, C); beta = rng.standard_normal(C)
    y_ref, cache = O.layernorm_fwd(x, gamma, beta)
    xd, gd = dev(x), dev(gamma); y, mean, rstd = zeros(R, C), zeros(R), zeros(R)
    L.call("sar_layernorm_fwd_f32", xd, gd, dev(beta), y, mean, rstd, R, C, 1e-14)
    check(y, y_ref, name="ln fwd")
    dy = rng.standard_normal((R, C)); dx_ref, dg_ref, db_ref = O.layernorm_bwd(dy, cache)
    dx, dg, db = zeros(R, C), zeros(C), zeros(C)
    L.call("sar_layernorm_bwd_f32", dev(dy), xd, gd, mean, rstd, dx, dg, db, R, C, 0)
    check(dx, dx_ref, name="ln dx"); check(dg, dg_ref, name="ln dgamma"); check(db, db_ref, name="ln dbeta")
    L.call("sar_layernorm_bwd_f32", dev(dy), xd, gd, mean, rstd, dx, dg, db, R, C, 1)
    check(dx, 2 * dx_ref, name="ln dx accumulate")
    # fused with the tanh in front of the LayerNorm (DS(..., 'tanh') -> LN): x = tanh(z), result = d/dz
    xt = np.tanh(x); yt_ref, cache_t = O.layernorm_fwd(xt, gamma, beta)
    xtd = dev(xt)
    L.call("sar_layernorm_fwd_f32", xtd, gd, dev(beta), y, mean, rstd, R, C, 1e-14)
    dxt_ref, dgt_ref, dbt_ref = O.layernorm_bwd(dy, cache_t)
    dx, dg, db = zeros(R, C), zeros(C), zeros(C)
    L.call("sar_layernorm_bwd_act_f32", dev(dy), xtd, gd, mean, rstd, dx, dg, db, R, C, 0, 1)
    check(dx, dxt_ref * (1 - xt * xt), name="ln+tanh dz"); check(dg, dgt_ref, name="ln+tanh dgamma"); check(db, dbt_ref, name="ln+tanh dbeta")


@pytest.mark.parametrize("R,K", [(832, 768), (832, 512), (2496, 256), (5, 64), (16, 128), (2496 + 7, 512)])
def test_dense_tanh_layernorm_one_launch(L, R, K):
    """sar_dense_tanh_ln_f32 (round 6; DS(256, 'tanh') -> LN, model.py:253-254 / 265-266 / 279-280): tanh's output, the normalised rows and the
    saved mean / rstd against the float64 oracle, at the step's shapes (832 x 768, 832 x 512), the CTC configurations' 2 496 rows and ragged
    row counts (the last workgroup's rows past R are read clamped and never stored); the two launches it replaces give the same tensors to fp32
    rounding; shapes it does not take return SAR_EINVAL and launch nothing."""
    rng = np.random.default_rng(R + K)
    N = 256
    x = rng.standard_normal((R, K)); W = rng.standard_normal((K, N)) / np.sqrt(K); b = 0.1 * rng.standard_normal(N)
    gamma = rng.uniform(0.5, 1.5, N); beta = rng.standard_normal(N)
    a_ref = np.tanh(x @ W + b); y_ref, cache = O.layernorm_fwd(a_ref, gamma, beta)
    xd, Wd, bd, gd, bed = dev(x), dev(W), dev(b), dev(gamma), dev(beta)
    a, y = torch.full((R + 1, N), 7.0, device="cuda"), torch.full((R + 1, N), 7.0, device="cuda")
    mean, rstd = zeros(R), zeros(R)
    old = L.get_gemm_math()
    L.set_gemm_math("bf16x3")
    try:
        L.call("sar_dense_tanh_ln_f32", xd, Wd, bd, gd, bed, a, y, mean, rstd, R, K, N, 1e-14)
        torch.cuda.synchronize()
        assert float((a[R] - 7.0).abs().max()) == 0.0 and float((y[R] - 7.0).abs().max()) == 0.0      # nothing past row R
        # (split precision drops lo x lo and rounds lo: ~2^-16 per product, ~2e-5 of a unit-variance sum; the LayerNorm divides by a deviation of ~0.6)
        check(a[:R], a_ref, 1e-4, name="tanh(x W + b)"); check(y[:R], y_ref, 3e-4, name="LN(tanh(x W + b))")
        mu_ref = a_ref.mean(1); rs_ref = 1.0 / np.sqrt(a_ref.var(1) + 1e-14)
        check(mean, mu_ref, 2e-5, name="mean", scale=1.0); check(rstd, rs_ref, 1e-4, name="rstd")
        # the two launches
        a2, y2, m2, r2 = zeros(R, N), zeros(R, N), zeros(R), zeros(R)
        L.call("sar_gemm_f32", xd, Wd, a2, bd, R, N, K, K, N, N, 1, 0, 1, 0, 1)
        L.call("sar_layernorm_fwd_f32", a2, gd, bed, y2, m2, r2, R, N, 1e-14)
        check(a[:R], a2.cpu().numpy(), 1e-4, name="against sar_gemm_f32"); check(y[:R], y2.cpu().numpy(), 3e-4, name="against sar_layernorm_fwd_f32")
        # not this kernel's shapes / arithmetic
        with pytest.raises(RuntimeError, match="status -1"):
            L.call("sar_dense_tanh_ln_f32", xd, Wd, bd, gd, bed, a, y, mean, rstd, R, K, 128, 1e-14)
        with pytest.raises(RuntimeError, match="status -1"):
            L.call("sar_dense_tanh_ln_f32", xd, Wd, bd, gd, bed, a, y, mean, rstd, R, K - 32, N, 1e-14)
        L.set_gemm_math("fp32")
        with pytest.raises(RuntimeError, match="status -1"):
            L.call("sar_dense_tanh_ln_f32", xd, Wd, bd, gd, bed, a, y, mean, rstd, R, K, N, 1e-14)
    finally:
        L.set_gemm_math(old)


# ------------------------------------------------------------------ BiGRU
@pytest.mark.parametrize("B,Lq,I,H", [(5, 7, 12, 64), (64, 39, 256, 256), (3, 12, 40, 20)])
def test_bigru(L, B, Lq, I, H):
    rng = np.random.default_rng(B * Lq)
    x = rng.standard_normal((B, Lq, I))
    mk = lambda: (rng.standard_normal((I, 3 * H)) / np.sqrt(I), rng.standard_normal((H, 3 * H)) / np.sqrt(H), rng.standard_normal(6 * H) * 0.1)
    pf, pb = mk(), mk()
    y_ref, cache = O.bigru_fwd(x, pf, pb, True)
    xd = dev(x)
    xa = [zeros(B * Lq, 3 * H), zeros(B * Lq, 3 * H)]
    W = [dev(pf[0]), dev(pb[0])]; U = [dev(pf[1]), dev(pb[1])]; bias = [dev(pf[2]), dev(pb[2])]
    for d in range(2):
        L.call("sar_gemm_f32", xd, W[d], xa[d], bias[d], B * Lq, 3 * H, I, I, 3 * H, 3 * H, 1, 0, 0, 0, 1)
    out = zeros(B, Lq, 2 * H); gates = zeros(2, B, Lq, 4, H); hprev = zeros(2, B, Lq, H)
    L.call("sar_bigru_fwd_f32", xa[0], xa[1], U[0], U[1], bias[0][3 * H:], bias[1][3 * H:], out, gates, hprev, B, Lq, H)
    check(out, y_ref, name="bigru fwd")
    dy = rng.standard_normal(y_ref.shape)
    dx_ref, gf, gb = O.bigru_bwd(dy, cache)
    UT = [zeros(3 * H, H), zeros(3 * H, H)]
    for d in range(2):
        L.call("sar_transpose_f32", U[d], UT[d], H, 3 * H)
    check(UT[0], pf[1].T, 1e-7, "transpose")
    dai = zeros(2, B, Lq, 3 * H); dah = zeros(2, B, Lq, 3 * H)
    L.call("sar_bigru_bwd_f32", dev(dy), gates, hprev, UT[0], UT[1], dai, dah, B, Lq, H)
    dx = zeros(B * Lq, I)
    for d, g in ((0, gf), (1, gb)):
        L.call("sar_gemm_f32", dai[d], W[d], dx, None, B * Lq, I, 3 * H, 3 * H, 3 * H, I, 1, 1, 0, d, 1)      # dx (+)= dai W^T
        dW = zeros(I, 3 * H); dU = zeros(H, 3 * H); db = zeros(6 * H)
        L.call("sar_gemm_f32", xd, dai[d], dW, None, I, 3 * H, B * Lq, I, 3 * H, 3 * H, 0, 0, 0, 0, 1)          # x^T dai
        L.call("sar_gemm_f32", hprev[d], dah[d], dU, None, H, 3 * H, B * Lq, H, 3 * H, 3 * H, 0, 0, 0, 0, 1)    # hprev^T dah
        L.call("sar_colsum_f32", dai[d], db[:3 * H], B * Lq, 3 * H); L.call("sar_colsum_f32", dah[d], db[3 * H:], B * Lq, 3 * H)
        check(dW, g[0], name="gru dW"); check(dU, g[1], name="gru dU"); check(db, g[2], name="gru db")
    check(dx.view(B, Lq, I), dx_ref, name="bigru dx")
    # seq=False helper
    yl = zeros(B, 2 * H); L.call("sar_bigru_last_fwd_f32", out, yl, B, Lq, H)
    check(yl, O.bigru_fwd(x, pf, pb, False)[0], name="bigru last")
    dl = rng.standard_normal((B, 2 * H)); dout = zeros(B, Lq, 2 * H)
    L.call("sar_bigru_last_bwd_f32", dev(dl), dout, B, Lq, H)
    ref = np.zeros((B, Lq, 2 * H)); ref[:, -1, :H] = dl[:, :H]; ref[:, 0, H:] = dl[:, H:]
    check(dout, ref, 1e-7, "bigru last bwd")


def _gru_bf16_emulation(xa, U, bh, dy_dir, reverse):
    """One direction of the bf16-MFMA recurrence in float64, rounding to bf16 exactly where gru_mfma.hip does: U once, h before
    every recurrent matmul, the recurrent-side pre-activation gradients before the BPTT matmul.  xa (B, L, 3H) holds x*W + b_i.
    Returns (out (B, L, H), dai, dah)."""
    bf = lambda a: torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).double().numpy()
    B, Lq, H3 = xa.shape; H = H3 // 3
    Ub = bf(U); sig = lambda a: 1.0 / (1.0 + np.exp(-a))
    order = range(Lq - 1, -1, -1) if reverse else range(Lq)
    h = np.zeros((B, H)); out = np.zeros((B, Lq, H)); saved = {}
    for t in order:
        G = bf(h) @ Ub
        z = sig(xa[:, t, :H] + G[:, :H] + bh[:H]); r = sig(xa[:, t, H:2 * H] + G[:, H:2 * H] + bh[H:2 * H]); q = G[:, 2 * H:] + bh[2 * H:]
        hh = np.tanh(xa[:, t, 2 * H:] + r * q)
        saved[t] = (z, r, hh, q, h)
        h = z * h + (1 - z) * hh
        out[:, t] = h
    dai = np.zeros((B, Lq, H3)); dah = np.zeros((B, Lq, H3)); dh = np.zeros((B, H))
    for t in reversed(list(order)):
        z, r, hh, q, hp = saved[t]
        g = dh + dy_dir[:, t]
        da_h = g * (1 - z) * (1 - hh * hh); da_z = g * (hp - hh) * z * (1 - z); da_r = da_h * q * r * (1 - r)
        dai[:, t] = np.concatenate([da_z, da_r, da_h], 1); a = np.concatenate([da_z, da_r, da_h * r], 1); dah[:, t] = a
        dh = g * z + bf(a) @ Ub.T
    return out, dai, dah


@pytest.mark.parametrize("B,Lq", [(64, 39), (5, 7), (19, 3), (8, 5), (520, 2)])   # four-row kernels: all rows live / ragged; 520: eight-row workgroups
def test_bigru_bf16_mfma(L, B, Lq):
    """gru_mfma.hip (the "bf16" mode at hidden_dim 256): against a float64 emulation that rounds the MFMA operands where the
    kernels do (wiring errors are O(1); agreement is limited by last-bit rounding flips of h), and loosely against the exact
    float64 GRU of the oracle (what bf16 operands cost)."""
    H, I = 256, 24
    rng = np.random.default_rng(B * Lq + 1)
    x = rng.standard_normal((B, Lq, I))
    mk = lambda: (rng.standard_normal((I, 3 * H)) / np.sqrt(I), rng.standard_normal((H, 3 * H)) / np.sqrt(H), rng.standard_normal(6 * H) * 0.1)
    P = [mk(), mk()]
    dy = rng.standard_normal((B, Lq, 2 * H))
    xa_np = [x @ p[0] + p[2][:3 * H] for p in P]
    xa = [dev(a.reshape(B * Lq, 3 * H)) for a in xa_np]
    U = [dev(p[1]) for p in P]; bh = [dev(p[2][3 * H:]) for p in P]
    pack = torch.zeros(2 * 2 * 3 * H * H, dtype=torch.bfloat16, device="cuda")
    L.call("sar_bigru_pack_bf16", U[0], U[1], pack, H)
    out = zeros(B, Lq, 2 * H); gates = zeros(2, B, Lq, 4, H); hprev = zeros(2, B, Lq, H)
    L.call("sar_bigru_fwd_bf16", xa[0], xa[1], pack, bh[0], bh[1], out, gates, hprev, B, Lq, H)
    dai = zeros(2, B, Lq, 3 * H); dah = zeros(2, B, Lq, 3 * H)
    L.call("sar_bigru_bwd_bf16", dev(dy), gates, hprev, pack, dai, dah, B, Lq, H)
    y_exact, cache = O.bigru_fwd(x, P[0], P[1], True)
    for d in range(2):
        o, ai, ah = _gru_bf16_emulation(xa_np[d], P[d][1], P[d][2][3 * H:], dy[:, :, d * H:(d + 1) * H], reverse=bool(d))
        check(out[:, :, d * H:(d + 1) * H], o, 3e-3, "bf16 gru out, direction %d" % d)
        check(dai[d], ai, 3e-3, "bf16 gru dai, direction %d" % d); check(dah[d], ah, 3e-3, "bf16 gru dah, direction %d" % d)
        hp = np.zeros_like(o)
        if d == 0: hp[:, 1:] = o[:, :-1]
        else: hp[:, :-1] = o[:, 1:]
        check(hprev[d], hp, 3e-3, "bf16 gru hprev, direction %d" % d)
    check(out, y_exact, 3e-2, "bf16 gru against the exact recurrence")


def test_zero_and_copy_ranges(L):
    """sar_zero_ranges / sar_copy_ranges: many buffers in one launch from a device table (the step-start zeroing and the device batch's
    input copies), sizes that are and are not multiples of four words, untouched neighbours."""
    sizes = [1, 3, 4, 5, 1024, 4099, 1 << 20, 2_621_447]             # (the last: past one sweep of the copy's four-loads-in-flight loop, ragged end)
    bufs = [torch.full((n + 8,), 7.0, device="cuda") for n in sizes]
    tab = torch.tensor([[b.data_ptr(), n] for b, n in zip(bufs, sizes)], dtype=torch.int64, device="cuda")
    L.call("sar_zero_ranges", tab, len(sizes))
    for b, n in zip(bufs, sizes):
        assert float(b[:n].abs().max()) == 0.0 and bool((b[n:] == 7.0).all())
    src = [torch.randn(n, device="cuda") for n in sizes]
    dst = [torch.full((n + 8,), 7.0, device="cuda") for n in sizes]
    tab = torch.tensor([[d.data_ptr(), s_.data_ptr(), n] for d, s_, n in zip(dst, src, sizes)], dtype=torch.int64, device="cuda")
    L.call("sar_copy_ranges", tab, len(sizes))
    for d, s_, n in zip(dst, src, sizes):
        assert torch.equal(d[:n], s_) and bool((d[n:] == 7.0).all())
    xi = torch.arange(10, dtype=torch.int32, device="cuda"); yi = torch.zeros(12, dtype=torch.int32, device="cuda")   # words, whatever the dtype
    L.call("sar_copy_ranges", torch.tensor([[yi.data_ptr(), xi.data_ptr(), 10]], dtype=torch.int64, device="cuda"), 1)
    assert torch.equal(yi[:10], xi) and int(yi[10:].abs().max()) == 0


# ------------------------------------------------------------------ CTC
def _ctc_inputs(B, Lq, V, maxU, rng, Umax):
    p = O.softmax(rng.standard_normal((B, Lq, V)) * 2)
    lab = np.full((B, maxU), 2.0, np.float32); ll = np.zeros((B, 1), np.int32)
    for b in range(B):
        u = int(rng.integers(1, Umax + 1)); ll[b] = u
        lab[b, :u] = rng.integers(0, V - 1, u)
    il = np.full((B, 1), Lq, np.int32)
    return p, lab, il, ll


# (3, 300, 20, 140, 135): S = 2 U + 1 up to 271 states -- more than the 256 threads the recursion keeps (the waves that only serve
# the parallel phases leave in front of the recursion's barriers); (3, 1, 7, 2, 1): a single frame
# (4, 114, 1000, 72, 40): the reference's shipped configuration (1200 frames): 66 / 132 KB of LDS staging, past the default 64 KB limit (round 4);
# (3, 189, ...): forward staged in LDS, backward (219 KB) on the workspace; (3, 300, ...): both on the workspace
@pytest.mark.parametrize("B,Lq,V,maxU,Umax", [(4, 12, 9, 8, 4), (64, 39, 1000, 72, 16), (4, 114, 1000, 72, 40), (3, 189, 50, 72, 60), (3, 300, 20, 140, 135), (3, 1, 7, 2, 1)])
def test_ctc(L, B, Lq, V, maxU, Umax):
    rng = np.random.default_rng(V)
    p, lab, il, ll = _ctc_inputs(B, Lq, V, maxU, rng, Umax)
    lab[0, 1] = lab[0, 0]; ll[0] = max(ll[0], 2)            # a repeated label
    il[1] = max(Lq - 3, 1)                                   # a shorter input
    if B > 2:
        ll[2] = min(maxU, Lq + 1); lab[2, :ll[2, 0]] = rng.integers(0, V - 1, ll[2, 0])   # infeasible: U > T
    loss_ref, grad_ref = O.ctc_batch_cost_fwd_bwd(p, lab, il, ll)
    pd, labd, ild, lld = dev(p), dev(lab), dev(il.ravel(), torch.int32), dev(ll.ravel(), torch.int32)
    wsb = L.ctc_workspace_bytes(B, Lq, maxU); ws = zeros(wsb // 4)
    loss = zeros(B)
    L.call("sar_ctc_batch_cost_fwd_f32", pd, labd, ild, lld, loss, ws, wsb, B, Lq, V, maxU)
    got = loss.cpu().numpy(); fin = np.isfinite(loss_ref[:, 0])
    assert np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], loss_ref[fin, 0], rtol=1e-4, atol=1e-4)
    dloss = rng.uniform(0.5, 1.5, B); dp = torch.full((B, Lq, V), 7.0, device="cuda", dtype=torch.float32)
    L.call("sar_ctc_batch_cost_bwd_f32", pd, labd, ild, lld, loss, dev(dloss), ws, wsb, dp, B, Lq, V, maxU)
    check(dp, grad_ref * dloss[:, None, None], 5e-4, "ctc grad")
    dec = zeros(B, Lq, dtype=torch.int32); dl = zeros(B, dtype=torch.int32)
    L.call("sar_ctc_greedy_decode_f32", pd, ild, dec, dl, B, Lq, V)
    ref = O.ctc_greedy_decode(p, il)
    for b in range(B):
        assert dec[b, :int(dl[b])].cpu().tolist() == ref[b]


# ------------------------------------------------------------------ VLAD / heads
@pytest.mark.parametrize("B,Lq,D,K,G", [(3, 6, 20, 4, 0), (64, 39, 256, 32, 2), (4, 189, 256, 64, 2), (5, 12, 64, 8, 2)])
def test_vlad(L, B, Lq, D, K, G):
    rng = np.random.default_rng(K + D)
    f = rng.standard_normal((B, 1, Lq, D)); s = rng.standard_normal((B, 1, Lq, K + G)) * 2; c = rng.standard_normal((K + G, D))
    y_ref, cache = O.vlad_fwd(f, s, c, K, "gvlad" if G else "vlad")
    fd, cd = dev(f), dev(c)
    out, A, inv = zeros(B, K * D), zeros(B, Lq, K + G), zeros(B, K)
    L.call("sar_vlad_fwd_f32", fd, dev(s), cd, out, A, inv, B, Lq, D, K, G)
    check(out, y_ref, name="vlad fwd")
    dy = rng.standard_normal(y_ref.shape)
    df_ref, ds_ref, dc_ref = O.vlad_bwd(dy, cache)
    df, ds, dc = zeros(B, Lq, D), zeros(B, Lq, K + G), zeros(K + G, D)
    L.call("sar_vlad_bwd_f32", dev(dy), fd, A, cd, out, inv, df, ds, dc, B, Lq, D, K, G, 0)
    check(df, df_ref.reshape(B, Lq, D), name="vlad dfeat"); check(ds, ds_ref.reshape(B, Lq, K + G), name="vlad dscore")
    check(dc, dc_ref, name="vlad dcenters")


@pytest.mark.parametrize("mode,name,m", [(0, "sphereface", 1.35), (1, "cosface", 0.3), (2, "arcface", 0.3)])
def test_margin_heads(L, mode, name, m):
    rng = np.random.default_rng(mode)
    B, D, A = 64, 256, 8
    x = rng.standard_normal((B, D)); W = rng.standard_normal((D, A)) * 0.1; y = np.eye(A)[rng.integers(0, A, B)]
    p_ref, cache = O.margin_head_fwd(x, y, W, name, 30.0, m)
    xd, yd, Wd = dev(x), dev(y), dev(W)
    p, cosv, ix, iw = zeros(B, A), zeros(B, A), zeros(B), zeros(A)
    L.call("sar_margin_head_fwd_f32", xd, yd, Wd, p, cosv, ix, iw, B, D, A, mode, 30.0, m)
    check(p, p_ref, name=name + " fwd")
    dp = rng.standard_normal((B, A))
    dx_ref, dW_ref = O.margin_head_bwd(dp, cache)
    dx, dW, ws = zeros(B, D), zeros(D, A), zeros(D, A)
    L.call("sar_margin_head_bwd_f32", dev(dp), p, cosv, xd, yd, Wd, ix, iw, dx, dW, ws, B, D, A, mode, 30.0, m, 0)
    check(dx, dx_ref, 1e-3, name + " dx"); check(dW, dW_ref, 1e-3, name + " dW")


def test_circle_cce_softmax_misc(L):
    rng = np.random.default_rng(11)
    B, A = 64, 8
    s = np.tanh(rng.standard_normal((B, A))); y = np.eye(A)[rng.integers(0, A, B)]
    loss_ref, ds_ref = O.circle_loss_fwd_bwd(y, s, 256.0, 0.3)
    sd, yd = dev(s), dev(y); loss = zeros(B)
    L.call("sar_circle_loss_fwd_f32", yd, sd, loss, B, A, 256.0, 0.3)
    check(loss, loss_ref, name="circle fwd")
    g = rng.uniform(0.5, 2, B); ds = zeros(B, A)
    L.call("sar_circle_loss_bwd_f32", yd, sd, dev(g), ds, B, A, 256.0, 0.3)
    check(ds, ds_ref * g[:, None], name="circle bwd")
    p = O.softmax(rng.standard_normal((B, A)) * 6); p[0] = [1 - 7e-9] + [1e-9] * 7
    l_ref, dp_ref = O.categorical_crossentropy_fwd_bwd(y, p)
    pd = dev(p); L.call("sar_cce_fwd_f32", yd, pd, loss, B, A)
    check(loss, l_ref, name="cce fwd")
    dp = zeros(B, A); L.call("sar_cce_bwd_f32", yd, pd, dev(g), dp, B, A)
    check(dp, dp_ref * g[:, None], name="cce bwd")
    for N in (8, 1000, 132, 1024, 1500):                          # 68 .. 1024 columns, multiples of 4: the one-wave-per-row kernels (round 4); others: a block per row
        z = rng.standard_normal((70, N)) * 3; zd = dev(z); q = zeros(70, N)
        L.call("sar_softmax_fwd_f32", zd, q, 70, N); check(q, O.softmax(z), name="softmax fwd")
        dq = rng.standard_normal((70, N)); sm = O.softmax(z); dz = zeros(70, N)
        L.call("sar_softmax_bwd_f32", dev(dq), q, dz, 70, N)
        check(dz, sm * (dq - (dq * sm).sum(-1, keepdims=True)), name="softmax bwd")
    x = rng.standard_normal((64, 256)); x[3] = 0
    v_ref, cache = O.l2_normalize_fwd(x, 1); v, inv = zeros(64, 256), zeros(64)
    L.call("sar_l2norm_rows_fwd_f32", dev(x), v, inv, 64, 256); check(v, v_ref, name="l2n fwd")
    dv = rng.standard_normal((64, 256)); dx = zeros(64, 256)
    L.call("sar_l2norm_rows_bwd_f32", dev(dv), v, inv, dx, 64, 256, 0)
    ref = O.l2_normalize_bwd(dv, cache); ref[3] = 0; got = dx.clone(); got[3] = 0      # row 3 is the clamped branch (1e6 scale)
    check(got, ref, name="l2n bwd")
    x3 = rng.standard_normal((5, 12, 16)); y2 = zeros(5, 16)
    L.call("sar_avgpool1d_fwd_f32", dev(x3), y2, 5, 12, 16); check(y2, x3.mean(1), name="avgpool")
    dy2 = rng.standard_normal((5, 16)); dx3 = zeros(5, 12, 16)
    L.call("sar_avgpool1d_bwd_f32", dev(dy2), dx3, 5, 12, 16, 0); check(dx3, O.global_avg_pool_bwd(dy2, 12), name="avgpool bwd")
    yv = dev(np.ones(1000)); L.call("sar_axpy_f32", dev(np.arange(1000.0)), yv, 0.5, 1000); check(yv, 1 + 0.5 * np.arange(1000.0), 1e-7)
    L.call("sar_fill_f32", yv, 3.0, 1000); check(yv, np.full(1000, 3.0), 1e-7)
    dz = zeros(100); yy = rng.standard_normal(100); gg = rng.standard_normal(100)
    L.call("sar_act_bwd_f32", dev(gg), dev(yy), dz, 100, 1); check(dz, gg * (1 - yy * yy), name="tanh bwd")
    L.call("sar_act_bwd_f32", dev(gg), dev(yy), dz, 100, 2); check(dz, gg * (yy > 0), name="relu bwd")


def test_adam_l2_unitnorm(L):
    rng = np.random.default_rng(5)
    n = 256 * 40
    p0 = rng.standard_normal(n); g = rng.standard_normal(n) * 0.1
    l2 = np.where(np.arange(40) % 2 == 0, 1e-4, 0.0)
    st = torch.zeros(8, dtype=torch.float32, device="cuda")
    st[:5] = torch.tensor([0.01, 2e-4, 0.9, 0.999, 1e-7]); st[7] = 0.5
    st.view(torch.int32)[6] = 0
    p, m, v = dev(p0), zeros(n), zeros(n); l2l = zeros(1)
    pr, mr, vr = p0.copy(), np.zeros(n), np.zeros(n)
    l2e = np.repeat(l2, 256)
    for it in range(3):
        l2l.zero_()
        L.call("sar_adam_prepare", st)
        L.call("sar_adam_l2_step_f32", p, dev(g), m, v, dev(l2), st, l2l, n)
        ref_l2 = (l2e * pr * pr).sum()
        pr, mr, vr = O.adam_step(pr, 0.5 * g + 2 * l2e * pr, mr, vr, it, lr=0.01)
        check(p, pr, 1e-5, "adam p step %d" % it); check(l2l, [ref_l2], 1e-4, "l2 loss")
    assert int(st.view(torch.int32)[6]) == 3
    # the one-launch form (round 3): prepare + step, the last block advances the iteration count; three further steps continue the run
    tick = torch.zeros(8, dtype=torch.int32, device="cuda")
    for it in range(3, 6):
        l2l.zero_()
        L.call("sar_adam_l2_update_f32", p, dev(g), m, v, dev(l2), st, tick, l2l, n)
        ref_l2 = (l2e * pr * pr).sum()
        pr, mr, vr = O.adam_step(pr, 0.5 * g + 2 * l2e * pr, mr, vr, it, lr=0.01)
        check(p, pr, 1e-5, "fused adam p step %d" % it); check(l2l, [ref_l2], 1e-4, "l2 loss")
        assert int(tick[0].item()) == 0 and int(tick[2].item()) == 0 and int(st.view(torch.int32)[6]) == it + 1
    # the loss-scaling guard (tick[1] = 1: the modes with fp16 operands; round 4): sar_grads_check_f32 flags a gradient with a non-finite
    # element and the update then SKIPS the whole step -- weights, moments, iteration count unchanged, tick[2] counts it, the flag is cleared
    tick[1] = 1
    gbad = g.copy(); gbad[[3, 700, 5000]] = [np.inf, -np.inf, np.nan]
    p_before, m_before, v_before = p.clone(), m.clone(), v.clone()
    l2l.zero_()
    gd = dev(gbad)
    L.call("sar_grads_check_f32", gd, tick, n)
    assert int(tick[3].item()) == 1
    L.call("sar_adam_l2_update_f32", p, gd, m, v, dev(l2), st, tick, l2l, n)
    assert torch.equal(p, p_before) and torch.equal(m, m_before) and torch.equal(v, v_before)
    assert int(tick[2].item()) == 1 and int(tick[3].item()) == 0 and int(tick[0].item()) == 0 and int(st.view(torch.int32)[6]) == 6
    assert int(tick[4].item()) == 1                                          # skipped steps in a row
    check(l2l, [(l2e * pr * pr).sum()], 1e-4, "l2 loss of a skipped step")       # still the penalty of the (unchanged) weights
    # a clean gradient behind it: the check leaves the flag alone and the step proceeds as step 6
    l2l.zero_()
    L.call("sar_grads_check_f32", dev(g), tick, n)
    assert int(tick[3].item()) == 0
    L.call("sar_adam_l2_update_f32", p, dev(g), m, v, dev(l2), st, tick, l2l, n)
    pr, mr, vr = O.adam_step(pr, 0.5 * g + 2 * l2e * pr, mr, vr, 6, lr=0.01)
    check(p, pr, 1e-5, "guarded adam, clean step"); assert int(tick[2].item()) == 1 and int(st.view(torch.int32)[6]) == 7
    assert int(tick[4].item()) == 0                                          # ... reset by the applied step
    w = rng.standard_normal((256, 8)); wd = dev(w)
    L.call("sar_unit_norm_f32", wd, 256, 8); check(wd, O.unit_norm(w), 1e-5, "unit_norm")
