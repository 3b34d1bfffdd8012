"""CPU tests pinning the oracle (oracle/ops_np.py) against independent code:
torch-CPU built-ins (F.conv2d, nn.GRU, F.ctc_loss, F.layer_norm, F.batch_norm), torch autograd
for every analytic backward, and hand-derived known-answer tests."""
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops_np as O
from oracle import model_torch as MT

RNG = np.random.default_rng(7)


def T(a, rg=True):
    return torch.tensor(np.asarray(a, np.float64), requires_grad=rg)


def close(a, b, tol=1e-9):
    a = np.asarray(a); b = b.detach().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


@pytest.mark.parametrize("h,w,k,s,pad", [(10, 8, 3, 1, "same"), (10, 8, 3, 2, "same"), (11, 5, 3, 2, "same"),
                                          (20, 16, 7, 2, "same"), (13, 3, 1, 2, "valid"), (10, 8, 1, 1, "valid")])
def test_conv2d_vs_torch(h, w, k, s, pad):
    x = RNG.standard_normal((2, h, w, 3)); wt = RNG.standard_normal((k, k, 3, 4)); b = RNG.standard_normal(4)
    y, cache = O.conv2d_fwd(x, wt, b, (s, s), pad)
    xt, wtt, bt = T(x), T(wt), T(b)
    yt = MT.conv2d(xt, wtt, bt, s, pad)
    close(y, yt)
    dy = RNG.standard_normal(y.shape)
    yt.backward(torch.tensor(dy))
    dx, dw, db = O.conv2d_bwd(dy, cache)
    close(dx, xt.grad); close(dw, wtt.grad); close(db, bt.grad)


def test_same_padding_rule():
    # 7x7/2 on 400 -> pads (2,3); 3x3/2 on even n -> (0,1); on odd n -> (1,1)
    assert O.same_pads(400, 7, 2) == (200, 2, 3)
    assert O.same_pads(100, 3, 2) == (50, 0, 1)
    assert O.same_pads(25, 3, 2) == (13, 1, 1)
    assert O.same_pads(80, 3, 1) == (80, 1, 1)
    assert O.cal_descriptors(1200, 80) == 114 and O.cal_descriptors(400, 80) == 39
    assert O.cal_descriptors(100, 80) == 12 and O.cal_descriptors(2000, 80) == 189


@pytest.mark.parametrize("h,w", [(10, 8), (11, 5), (7, 7)])
def test_maxpool(h, w):
    x = RNG.standard_normal((2, h, w, 3))
    y, cache = O.maxpool_fwd(x)
    xt = T(x); yt = MT.maxpool(xt)
    close(y, yt)
    dy = RNG.standard_normal(y.shape); yt.backward(torch.tensor(dy))
    close(O.maxpool_bwd(dy, cache), xt.grad)


@pytest.mark.parametrize("shape", [(4, 5, 3, 6), (8, 6)])
def test_batchnorm(shape):
    x = RNG.standard_normal(shape) * 2 + 1; g = RNG.standard_normal(shape[-1]); b = RNG.standard_normal(shape[-1])
    y, cache = O.batchnorm_fwd(x, g, b)
    xt, gt, bt = T(x), T(g), T(b)
    # independent: F.batch_norm in training mode on channels-first view
    perm = (0, len(shape) - 1) + tuple(range(1, len(shape) - 1))
    inv = tuple(np.argsort(perm))
    yt = F.batch_norm(xt.permute(perm), None, None, gt, bt, True, 0.0, 1e-3).permute(inv)
    close(y, yt)
    dy = RNG.standard_normal(shape); yt.backward(torch.tensor(dy))
    dx, dg, db = O.batchnorm_bwd(dy, cache)
    close(dx, xt.grad); close(dg, gt.grad); close(db, bt.grad)


def test_layernorm():
    x = RNG.standard_normal((3, 5, 16)); g = RNG.standard_normal(16); b = RNG.standard_normal(16)
    y, cache = O.layernorm_fwd(x, g, b)
    xt, gt, bt = T(x), T(g), T(b)
    yt = F.layer_norm(xt, (16,), gt, bt, eps=1e-14)
    close(y, yt)
    dy = RNG.standard_normal(x.shape); yt.backward(torch.tensor(dy))
    dx, dg, db = O.layernorm_bwd(dy, cache)
    close(dx, xt.grad, 1e-8); close(dg, gt.grad); close(db, bt.grad)


@pytest.mark.parametrize("act", [None, "tanh", "relu", "softmax"])
def test_dense(act):
    x = RNG.standard_normal((3, 4, 6)); w = RNG.standard_normal((6, 5)); b = RNG.standard_normal(5)
    y, cache = O.dense_fwd(x, w, b, act)
    xt, wt, bt = T(x), T(w), T(b)
    z = xt @ wt + bt
    yt = {None: z, "tanh": torch.tanh(z), "relu": torch.relu(z), "softmax": torch.softmax(z, -1)}[act]
    close(y, yt)
    dy = RNG.standard_normal(y.shape); yt.backward(torch.tensor(dy))
    dx, dw, db = O.dense_bwd(dy, cache)
    close(dx, xt.grad); close(dw, wt.grad); close(db, bt.grad)


def _keras_to_torch_gru(k, rk, bias, H):
    """Keras z,r,h packing -> torch nn.GRU r,z,n packing."""
    def reorder(m):
        z, r, h = m[..., :H], m[..., H:2 * H], m[..., 2 * H:]
        return np.concatenate([r, z, h], -1)
    return reorder(k).T, reorder(rk).T, reorder(bias[:3 * H]), reorder(bias[3 * H:])


@pytest.mark.parametrize("seq", [True, False])
def test_bigru_vs_torch_nn_gru(seq):
    B, L, I, H = 3, 7, 5, 4
    x = RNG.standard_normal((B, L, I))
    pf = (RNG.standard_normal((I, 3 * H)) * .5, RNG.standard_normal((H, 3 * H)) * .5, RNG.standard_normal(6 * H) * .5)
    pb = (RNG.standard_normal((I, 3 * H)) * .5, RNG.standard_normal((H, 3 * H)) * .5, RNG.standard_normal(6 * H) * .5)
    y, cache = O.bigru_fwd(x, pf, pb, seq)
    gru = torch.nn.GRU(I, H, batch_first=True, bidirectional=True).double()
    with torch.no_grad():
        for sfx, p in (("", pf), ("_reverse", pb)):
            wi, wh, bi, bh = _keras_to_torch_gru(*p, H)
            getattr(gru, "weight_ih_l0" + sfx).copy_(torch.tensor(wi)); getattr(gru, "weight_hh_l0" + sfx).copy_(torch.tensor(wh))
            getattr(gru, "bias_ih_l0" + sfx).copy_(torch.tensor(bi)); getattr(gru, "bias_hh_l0" + sfx).copy_(torch.tensor(bh))
    xt = T(x)
    out, hn = gru(xt)
    yt = out if seq else torch.cat([hn[0], hn[1]], -1)
    close(y, yt)
    dy = RNG.standard_normal(y.shape); yt.backward(torch.tensor(dy))
    dx, gf, gb = O.bigru_bwd(dy, cache)
    close(dx, xt.grad)
    for sfx, g, p in (("", gf, pf), ("_reverse", gb, pb)):
        wi, wh, bi, bh = _keras_to_torch_gru(g[0], g[1], g[2], H)
        close(wi, getattr(gru, "weight_ih_l0" + sfx).grad); close(wh, getattr(gru, "weight_hh_l0" + sfx).grad)
        close(bi, getattr(gru, "bias_ih_l0" + sfx).grad); close(bh, getattr(gru, "bias_hh_l0" + sfx).grad)
    # and the explicit torch loop used by model_torch
    P = {"g/forward/kernel": T(pf[0]), "g/forward/recurrent_kernel": T(pf[1]), "g/forward/bias": T(pf[2]),
         "g/backward/kernel": T(pb[0]), "g/backward/recurrent_kernel": T(pb[1]), "g/backward/bias": T(pb[2])}
    close(y, MT.bigru(T(x), P, "g", seq))


def _ctc_case(B, L, V, maxU, rng, repeats=False):
    z = rng.standard_normal((B, L, V)); p = O.softmax(z)
    lab = np.full((B, maxU), 2.0, np.float32); ll = np.zeros((B, 1), np.int32)
    for b in range(B):
        u = rng.integers(1, maxU + 1); ll[b] = u
        lab[b, :u] = rng.integers(0, V - 1, u)
        if repeats and u >= 2:
            lab[b, 1] = lab[b, 0]
    il = np.full((B, 1), L, np.int32)
    return p, lab, il, ll


@pytest.mark.parametrize("repeats", [False, True])
def test_ctc_vs_torch(repeats):
    rng = np.random.default_rng(3)
    p, lab, il, ll = _ctc_case(4, 12, 9, 4, rng, repeats)
    il[1] = 9                                                  # shorter input for one utterance
    loss, grad = O.ctc_batch_cost_fwd_bwd(p, lab, il, ll)
    pt = T(p)
    lt = MT.ctc_batch_cost(lab, pt, il, ll)
    close(loss, lt, 1e-8)
    lt.sum().backward()
    close(grad, pt.grad, 1e-8)
    assert np.all(grad[1, 9:] == 0)


def test_ctc_known_answer_two_frames():
    # T=2, one label 'a' (index 0), V=3, blank=2: paths a-, -a, aa  => P = pa0*pb1 + pb0*pa1 + pa0*pa1
    p = np.array([[[0.5, 0.2, 0.3], [0.1, 0.3, 0.6]]])
    pe = p + 1e-7; y = pe / pe.sum(-1, keepdims=True)
    expect = -math.log(y[0, 0, 0] * y[0, 1, 2] + y[0, 0, 2] * y[0, 1, 0] + y[0, 0, 0] * y[0, 1, 0])
    loss, _ = O.ctc_batch_cost_fwd_bwd(p, np.array([[0.0, 2.0]]), np.array([[2]]), np.array([[1]]))
    assert abs(loss[0, 0] - expect) < 1e-12
    # the +1e-7 detour matters: differs from plain -log P(p)
    plain = -math.log(p[0, 0, 0] * p[0, 1, 2] + p[0, 0, 2] * p[0, 1, 0] + p[0, 0, 0] * p[0, 1, 0])
    assert abs(loss[0, 0] - plain) > 1e-8


def test_ctc_infeasible_is_inf_and_fd():
    p = O.softmax(RNG.standard_normal((1, 2, 5)))
    loss, grad = O.ctc_batch_cost_fwd_bwd(p, np.array([[1.0, 1.0, 3.0]]), np.array([[2]]), np.array([[3]]))
    assert np.isinf(loss[0, 0]) and np.all(grad == 0)
    # finite differences on a feasible case
    rng = np.random.default_rng(5)
    p, lab, il, ll = _ctc_case(1, 6, 5, 2, rng)
    loss, grad = O.ctc_batch_cost_fwd_bwd(p, lab, il, ll)
    for (t, v) in [(0, 0), (3, 4), (5, 2)]:
        d = np.zeros_like(p); d[0, t, v] = 1e-6
        lp, _ = O.ctc_batch_cost_fwd_bwd(p + d, lab, il, ll); lm, _ = O.ctc_batch_cost_fwd_bwd(p - d, lab, il, ll)
        assert abs((lp - lm)[0, 0] / 2e-6 - grad[0, t, v]) < 1e-5


@pytest.mark.parametrize("mode,K,G", [("vlad", 4, 0), ("gvlad", 4, 2)])
def test_vlad(mode, K, G):
    B, L, D = 3, 6, 5
    f = RNG.standard_normal((B, 1, L, D)); s = RNG.standard_normal((B, 1, L, K + G)); c = RNG.standard_normal((K + G, D))
    y, cache = O.vlad_fwd(f, s, c, K, mode)
    ft, st, ct = T(f), T(s), T(c)
    # verbatim VLAD.py:33-48 in torch (materialises the broadcast tensor)
    A = torch.exp(st - st.max(-1, keepdim=True)[0]); A = A / A.sum(-1, keepdim=True)
    wr = A.unsqueeze(-1) * (ft.unsqueeze(-2) - ct)
    cr = wr.sum((1, 2))[:, :K]
    yt = MT.l2n(cr, -1).reshape(B, K * D)
    close(y, yt)
    dy = RNG.standard_normal(y.shape); yt.backward(torch.tensor(dy))
    df, ds, dc = O.vlad_bwd(dy, cache)
    close(df, ft.grad); close(ds, st.grad); close(dc, ct.grad)


def test_vlad_hard_assignment_kat():
    # one-hot assignments => per-cluster sum of (x - c_k), then unit norm
    f = np.array([[[[1.0, 0.0], [3.0, 0.0], [0.0, 2.0]]]]); s = np.array([[[[50.0, -50], [50, -50], [-50, 50]]]])
    c = np.array([[1.0, 1.0], [0.0, 0.0]])
    y, _ = O.vlad_fwd(f, s, c, 2, "vlad")
    v0 = np.array([1 - 1 + 3 - 1, 0 - 1 + 0 - 1.0]); v1 = np.array([0.0, 2.0])
    np.testing.assert_allclose(y[0], np.concatenate([v0 / np.linalg.norm(v0), v1 / np.linalg.norm(v1)]), atol=1e-12)


@pytest.mark.parametrize("mode,m", [("cosface", 0.3), ("arcface", 0.3), ("sphereface", 1.35)])
def test_margin_heads(mode, m):
    B, D, A = 5, 6, 4
    x = RNG.standard_normal((B, D)); W = RNG.standard_normal((D, A)); y = np.eye(A)[RNG.integers(0, A, B)]
    p, cache = O.margin_head_fwd(x, y, W, mode, 30.0, m)
    xt, Wt, yt = T(x), T(W), T(y, False)
    c = MT.l2n(xt, 1) @ MT.l2n(Wt, 0)
    if mode == "cosface":
        t = c - m
    else:
        th = torch.acos(torch.clamp(c, -1 + 1e-7, 1 - 1e-7))
        t = torch.cos(th + m) if mode == "arcface" else torch.cos(m * th)
    pt = torch.softmax((c * (1 - yt) + t * yt) * 30.0, -1)
    close(p, pt)
    dp = RNG.standard_normal(p.shape); pt.backward(torch.tensor(dp))
    dx, dW = O.margin_head_bwd(dp, cache)
    close(dx, xt.grad, 1e-8); close(dW, Wt.grad, 1e-8)


def test_arcface_kat():
    # x aligned with column 0 at angle 0.3 => target logit = s*cos(0.6)
    th = 0.3
    x = np.array([[math.cos(th), math.sin(th)]]); W = np.array([[1.0, 0.0], [0.0, 1.0]]); y = np.array([[1.0, 0.0]])
    p, _ = O.margin_head_fwd(x, y, W, "arcface", 30.0, 0.3)
    lg = np.array([30 * math.cos(0.6), 30 * math.sin(th)])
    np.testing.assert_allclose(p[0], np.exp(lg - lg.max()) / np.exp(lg - lg.max()).sum(), atol=1e-9)


def test_circle_and_cce():
    B, A = 6, 5
    s = np.tanh(RNG.standard_normal((B, A))); y = np.eye(A)[RNG.integers(0, A, B)]
    loss, ds = O.circle_loss_fwd_bwd(y, s, 256.0, 0.3)
    st = T(s); lt = MT.circle_loss(T(y, False), st, 256.0, 0.3)
    close(loss, lt, 1e-8); lt.sum().backward(); close(ds, st.grad, 1e-7)
    # KAT: at s_p = Delta_p and s_n = Delta_n all logits are zero -> loss = log(A)
    sk = np.where(y > 0, 0.7, 0.3); lk, _ = O.circle_loss_fwd_bwd(y, sk, 256.0, 0.3)
    np.testing.assert_allclose(lk, math.log(A), atol=1e-12)
    p = O.softmax(RNG.standard_normal((B, A)) * 8); p[0] = [1 - 4e-9, 1e-9, 1e-9, 1e-9, 1e-9]
    l2, dp = O.categorical_crossentropy_fwd_bwd(y, p)
    pt = T(p); l2t = MT.cce(T(y, False), pt)
    close(l2, l2t); l2t.sum().backward(); close(dp, pt.grad, 1e-8)


def test_l2norm_gap_adam_unitnorm():
    x = RNG.standard_normal((4, 7)); x[2] = 0
    v, cache = O.l2_normalize_fwd(x, 1)
    xt = T(x); vt = MT.l2n(xt, 1); close(v, vt)
    dv = RNG.standard_normal(x.shape); vt.backward(torch.tensor(dv)); close(O.l2_normalize_bwd(dv, cache), xt.grad)
    p = RNG.standard_normal(10); g = RNG.standard_normal(10); m = np.zeros(10); v_ = np.zeros(10)
    # step 0 of Keras Adam: lr_t = lr*sqrt(1-b2)/(1-b1); m=(1-b1)g; v=(1-b2)g^2
    p1, m1, v1 = O.adam_step(p, g, m, v_, 0, lr=0.01)
    lr_t = 0.01 * math.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(p1, p - lr_t * 0.1 * g / (np.sqrt(0.001 * g * g) + 1e-7), rtol=1e-12)
    p2, _, _ = O.adam_step(p1, g, m1, v1, 1, lr=0.01)
    assert np.all(np.isfinite(p2))
    w = RNG.standard_normal((5, 3)); np.testing.assert_allclose(np.linalg.norm(O.unit_norm(w), axis=0), 1.0, atol=1e-6)
    assert O.ctc_greedy_decode(np.eye(4)[[0, 0, 3, 0, 1, 1, 3]][None], [7]) == [[0, 0, 1]]


def test_whole_graph_inference_mode_and_bottleneck_head():
    """oracle/model_torch.py: (a) BatchNormalization moving averages follow m <- 0.99 m + 0.01 batch (2-D: Keras' sample variance var * n / (n - (1 + eps)));
    (b) with the moving averages set to the batch statistics the inference graph equals the training graph;
    (c) the bn_dim head adds y_disc_bn with weight 0.1 and always-CE loss (model.py:363-366)."""
    import torch
    from oracle import model_torch as MT
    cfg = MT.Cfg([64, 80, 1], disc_enable=True, res_type="res18", res_filters=16, hidden_dim=32, mto="avg",
                 metric_loss="cosface", bn_dim=2)
    net = MT.TorchSARNet(cfg, seed=1)
    b = MT.synthetic_batch(cfg, 5, seed=3)
    assert MT.loss_weights(cfg) == {"y_accent": 0.01, "y_disc": 1.0, "y_disc_bn": 0.1}
    out, total, parts, G = net.forward_backward(b)
    assert set(parts) == {"y_accent", "y_disc", "y_disc_bn", "l2"} and out["y_disc_bn"].shape == (5, 8)
    want = 0.01 * parts["y_accent"] + parts["y_disc"] + 0.1 * parts["y_disc_bn"] + parts["l2"]
    assert abs(float(total) - float(want)) < 1e-12
    # (a) 2-D BN (AR_BN4 on the bottleneck pre-activation)
    with torch.no_grad():
        P = net.P
        v = torch.relu(out["ar"] @ P["AR_BN_DS/kernel"] + P["AR_BN_DS/bias"])
        mu3 = v.mean(0); var3 = v.var(0, unbiased=False)
    np.testing.assert_allclose(net.state["AR_BN3/moving_mean"], 0.01 * mu3.numpy(), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(net.state["AR_BN3/moving_variance"], 0.99 + 0.01 * var3.numpy() * 5 / (5 - 1.001), rtol=1e-12)
    # (b) with momentum 0 the state becomes the batch statistics of this pass; the inference graph then reproduces the
    # training graph up to the n/(n-1) Bessel factor the 4-D layers store in moving_variance (n >= 5*2*3 here)
    st = dict(net.state)
    old_m = MT.BN_MOMENTUM
    MT.BN_MOMENTUM = 0.0
    try:
        trn = MT.forward(net.P, b, cfg, state=st, training=True)
    finally:
        MT.BN_MOMENTUM = old_m
    inf = MT.forward(net.P, b, cfg, state=st, training=False)
    # ... and the n / (n - (1 + eps)) sample-variance factor of the 2-D layers: 5 / 3.999 = 1.25 at this batch size of 5
    assert float((inf["y_accent"] - trn["y_accent"]).abs().max()) < 1.5e-1
    assert float((inf["y_accent"] - trn["y_accent"]).abs().max()) > 0          # the Bessel factor is really there
    assert np.isfinite(inf["y_disc_bn"].detach().numpy()).all()
