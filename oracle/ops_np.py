"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the SAR-Net train-step hot path.

numpy float64 restatement (forward + analytic backward) of every op the
reference composes on its hot path.  Each function cites the reference
file:line it follows (paths relative to /root/reference) and, where the
arithmetic lives in un-vendored Keras 2.2.4 / TF 1.13.1 /
keras-layer-normalization, the SURVEY.md Appendix-A row that pins the
semantics.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures,
and TensorFlow/Keras cannot be imported in the build container, so this
oracle is pinned only by (a) hand-derivable known-answer tests, (b) torch-CPU
built-ins as an independent second opinion (tests/test_oracle_*.py) and
(c) finite differences.  See DESIGN.md section "Oracle".

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  The product (aesrc2020_amd/) never does.
"""
import numpy as np

F64 = np.float64
K_EPS = 1e-7          # keras.backend.epsilon()
BN_EPS = 1e-3         # keras BatchNormalization default epsilon
LN_EPS = 1e-14        # keras_layer_normalization: K.epsilon()**2
L2N_EPS = 1e-12       # tf.nn.l2_normalize epsilon


# --------------------------------------------------------------------------
# padding helpers (TF "SAME" / "VALID") -- SURVEY App. A rows 1-3
# --------------------------------------------------------------------------
def same_pads(n, k, s):
    """TF SAME: out=ceil(n/s); total=max((out-1)s+k-n,0); before=total//2."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def valid_out(n, k, s):
    return (n - k) // s + 1


def _im2col(xp, kh, kw, sh, sw, ho, wo):
    b, _, _, c = xp.shape
    cols = np.empty((b, ho, wo, kh, kw, c), dtype=xp.dtype)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i, j, :] = xp[:, i:i + (ho - 1) * sh + 1:sh,
                                        j:j + (wo - 1) * sw + 1:sw, :]
    return cols


def conv2d_fwd(x, w, b, stride, padding):
    """Keras Conv2D, NHWC input, HWIO kernel (resnet.py:39-42,60-63,82-87,113-117;
    model.py:89-95).  padding in {"same","valid"}; stride (sh, sw)."""
    x = np.asarray(x, F64); w = np.asarray(w, F64)
    kh, kw, ci, co = w.shape
    sh, sw = stride
    bsz, h, wd, _ = x.shape
    if padding == "same":
        ho, pt, pb = same_pads(h, kh, sh)
        wo, pl, pr = same_pads(wd, kw, sw)
    else:
        ho, wo = valid_out(h, kh, sh), valid_out(wd, kw, sw)
        pt = pb = pl = pr = 0
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    cols = _im2col(xp, kh, kw, sh, sw, ho, wo)
    y = cols.reshape(-1, kh * kw * ci) @ w.reshape(-1, co)
    if b is not None:
        y = y + np.asarray(b, F64)
    cache = (x.shape, cols, w, (sh, sw), (pt, pb, pl, pr), (ho, wo), b is not None)
    return y.reshape(bsz, ho, wo, co), cache


def conv2d_bwd(dy, cache):
    xshape, cols, w, (sh, sw), (pt, pb, pl, pr), (ho, wo), has_b = cache
    kh, kw, ci, co = w.shape
    bsz, h, wd, _ = xshape
    dy2 = np.asarray(dy, F64).reshape(-1, co)
    dw = (cols.reshape(-1, kh * kw * ci).T @ dy2).reshape(w.shape)
    db = dy2.sum(0) if has_b else None
    dcols = (dy2 @ w.reshape(-1, co).T).reshape(bsz, ho, wo, kh, kw, ci)
    dxp = np.zeros((bsz, h + pt + pb, wd + pl + pr, ci), F64)
    for i in range(kh):
        for j in range(kw):
            dxp[:, i:i + (ho - 1) * sh + 1:sh, j:j + (wo - 1) * sw + 1:sw, :] += dcols[:, :, :, i, j, :]
    dx = dxp[:, pt:pt + h, pl:pl + wd, :]
    return dx, dw, db


def maxpool_fwd(x, k=3, s=2):
    """MaxPooling2D(pool 3x3, strides 2, padding 'same') -- resnet.py:174,192.
    TF SAME with -inf padding (App. A)."""
    x = np.asarray(x, F64)
    bsz, h, wd, c = x.shape
    ho, pt, pb = same_pads(h, k, s)
    wo, pl, pr = same_pads(wd, k, s)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
    cols = _im2col(xp, k, k, s, s, ho, wo).reshape(bsz, ho, wo, k * k, c)
    arg = cols.argmax(3)          # first max wins (matches a strict '>' scan)
    y = np.take_along_axis(cols, arg[:, :, :, None, :], 3)[:, :, :, 0, :]
    return y, (x.shape, arg, k, s, (pt, pb, pl, pr))


def maxpool_bwd(dy, cache):
    xshape, arg, k, s, (pt, pb, pl, pr) = cache
    bsz, h, wd, c = xshape
    _, ho, wo, _ = dy.shape
    dxp = np.zeros((bsz, h + pt + pb, wd + pl + pr, c), F64)
    bi, hi, wi, ci = np.meshgrid(np.arange(bsz), np.arange(ho), np.arange(wo), np.arange(c), indexing="ij")
    np.add.at(dxp, (bi, hi * s + arg // k, wi * s + arg % k, ci), dy)
    return dxp[:, pt:pt + h, pl:pl + wd, :]


# --------------------------------------------------------------------------
# normalisation
# --------------------------------------------------------------------------
def batchnorm_fwd(x, gamma, beta, eps=BN_EPS):
    """BatchNormalization() in training mode over all axes but the last
    (resnet.py:25; model.py:29-30,286,289).  Biased batch variance (App. A)."""
    x = np.asarray(x, F64)
    ax = tuple(range(x.ndim - 1))
    mu = x.mean(ax); var = x.var(ax)
    rstd = 1.0 / np.sqrt(var + eps)
    xh = (x - mu) * rstd
    return xh * gamma + beta, (xh, rstd, np.asarray(gamma, F64), ax, mu, var)


def batchnorm_bwd(dy, cache):
    xh, rstd, gamma, ax, _, _ = cache
    dy = np.asarray(dy, F64)
    n = np.prod([xh.shape[a] for a in ax])
    dgamma = (dy * xh).sum(ax); dbeta = dy.sum(ax)
    dx = gamma * rstd * (dy - dbeta / n - xh * dgamma / n)
    return dx, dgamma, dbeta


def relu_fwd(x):
    return np.maximum(x, 0.0)


def relu_bwd(dy, y):
    return dy * (y > 0)


def layernorm_fwd(x, gamma, beta, eps=LN_EPS):
    """keras_layer_normalization.LayerNormalization() (model.py:4,32-33):
    mean/var over last axis, y=(x-mu)/sqrt(var+eps)*gamma+beta, eps=K.epsilon()**2."""
    x = np.asarray(x, F64)
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xh = (x - mu) * rstd
    return xh * gamma + beta, (xh, rstd, np.asarray(gamma, F64))


def layernorm_bwd(dy, cache):
    xh, rstd, gamma = cache
    dy = np.asarray(dy, F64)
    red = tuple(range(dy.ndim - 1))
    dgamma = (dy * xh).sum(red); dbeta = dy.sum(red)
    g = dy * gamma
    dx = rstd * (g - g.mean(-1, keepdims=True) - xh * (g * xh).mean(-1, keepdims=True))
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------
# dense / activations
# --------------------------------------------------------------------------
def softmax(z, axis=-1):
    z = z - z.max(axis, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis, keepdims=True)


def dense_fwd(x, w, b, act=None):
    """Dense (model.py:35-42 DS; 163; 150).  act in {None,'tanh','relu','softmax'}."""
    x = np.asarray(x, F64)
    z = x @ np.asarray(w, F64)
    if b is not None:
        z = z + b
    if act is None:
        y = z
    elif act == "tanh":
        y = np.tanh(z)
    elif act == "relu":
        y = np.maximum(z, 0)
    elif act == "softmax":
        y = softmax(z)
    else:
        raise ValueError(act)
    return y, (x, np.asarray(w, F64), y, act, b is not None)


def dense_bwd(dy, cache):
    x, w, y, act, has_b = cache
    dy = np.asarray(dy, F64)
    if act is None:
        dz = dy
    elif act == "tanh":
        dz = dy * (1 - y * y)
    elif act == "relu":
        dz = dy * (y > 0)
    elif act == "softmax":
        dz = y * (dy - (dy * y).sum(-1, keepdims=True))
    x2 = x.reshape(-1, x.shape[-1]); dz2 = dz.reshape(-1, dz.shape[-1])
    dw = x2.T @ dz2
    db = dz2.sum(0) if has_b else None
    dx = dz @ w.T
    return dx, dw, db


def sigmoid(a):
    return 1.0 / (1.0 + np.exp(-a))


# --------------------------------------------------------------------------
# Bidirectional(CuDNNGRU) -- model.py:44-50; App. A row "CuDNNGRU"
# --------------------------------------------------------------------------
def gru_dir_fwd(x, kernel, rkernel, bias, reverse=False):
    """One direction.  kernel (in,3H), rkernel (H,3H), bias (6H)=[b_iz,b_ir,b_ih,b_hz,b_hr,b_hh];
    gate order z,r,h; reset-after: hh=tanh(xh + r*(hU_h+b_hh)); h'=z*h+(1-z)*hh; h0=0.
    Returns outputs aligned to input time (Keras Bidirectional re-reverses the backward layer)."""
    x = np.asarray(x, F64)
    bsz, L, _ = x.shape
    H = rkernel.shape[0]
    bi, bh = bias[:3 * H], bias[3 * H:]
    xa = x @ kernel + bi                                   # (B,L,3H)
    h = np.zeros((bsz, H), F64)
    out = np.zeros((bsz, L, H), F64)
    zs = np.zeros((bsz, L, H)); rs = np.zeros((bsz, L, H)); hhs = np.zeros((bsz, L, H)); qs = np.zeros((bsz, L, H))
    hprev = np.zeros((bsz, L, H))
    order = range(L - 1, -1, -1) if reverse else range(L)
    for t in order:
        ha = h @ rkernel + bh
        z = sigmoid(xa[:, t, :H] + ha[:, :H])
        r = sigmoid(xa[:, t, H:2 * H] + ha[:, H:2 * H])
        q = ha[:, 2 * H:]
        hh = np.tanh(xa[:, t, 2 * H:] + r * q)
        hprev[:, t] = h
        h = z * h + (1 - z) * hh
        out[:, t] = h; zs[:, t] = z; rs[:, t] = r; hhs[:, t] = hh; qs[:, t] = q
    return out, (x, kernel, rkernel, zs, rs, hhs, qs, hprev, reverse)


def gru_dir_bwd(dout, cache):
    x, kernel, rkernel, zs, rs, hhs, qs, hprev, reverse = cache
    bsz, L, H = zs.shape
    dout = np.asarray(dout, F64)
    dai = np.zeros((bsz, L, 3 * H)); dah = np.zeros((bsz, L, 3 * H))
    dh = np.zeros((bsz, H))
    order = range(L) if reverse else range(L - 1, -1, -1)      # reverse of processing order
    for t in order:
        dh = dh + dout[:, t]
        z, r, hh, q, hp = zs[:, t], rs[:, t], hhs[:, t], qs[:, t], hprev[:, t]
        dz = dh * (hp - hh); dhh = dh * (1 - z)
        da_h = dhh * (1 - hh * hh)
        da_z = dz * z * (1 - z)
        da_r = (da_h * q) * r * (1 - r)
        dai[:, t] = np.concatenate([da_z, da_r, da_h], 1)
        dah[:, t] = np.concatenate([da_z, da_r, da_h * r], 1)
        dh = dh * z + dah[:, t] @ rkernel.T
    x2 = x.reshape(-1, x.shape[-1])
    dkernel = x2.T @ dai.reshape(-1, 3 * H)
    drkernel = hprev.reshape(-1, H).T @ dah.reshape(-1, 3 * H)
    dbias = np.concatenate([dai.sum((0, 1)), dah.sum((0, 1))])
    dx = dai @ kernel.T
    return dx, dkernel, drkernel, dbias


def bigru_fwd(x, pf, pb, seq=True):
    """Bidirectional(CuDNNGRU, merge_mode='concat').  pf/pb = (kernel, rkernel, bias).
    seq=False returns [h_fwd(L-1), h_bwd(t=0)] (model.py:128)."""
    of, cf = gru_dir_fwd(x, *pf, reverse=False)
    ob, cb = gru_dir_fwd(x, *pb, reverse=True)
    if seq:
        y = np.concatenate([of, ob], -1)
    else:
        y = np.concatenate([of[:, -1], ob[:, 0]], -1)
    return y, (cf, cb, seq, of.shape)


def bigru_bwd(dy, cache):
    cf, cb, seq, oshape = cache
    H = oshape[-1]
    if seq:
        dof, dob = dy[..., :H], dy[..., H:]
    else:
        dof = np.zeros(oshape); dob = np.zeros(oshape)
        dof[:, -1] = dy[:, :H]; dob[:, 0] = dy[:, H:]
    dxf, *gf = gru_dir_bwd(dof, cf)
    dxb, *gb = gru_dir_bwd(dob, cb)
    return dxf + dxb, tuple(gf), tuple(gb)


# --------------------------------------------------------------------------
# K.ctc_batch_cost -- model.py:62-64; App. A row "K.ctc_batch_cost"
# --------------------------------------------------------------------------
def _logsumexp(a, b):
    m = np.maximum(a, b)
    with np.errstate(invalid="ignore"):
        r = m + np.log(np.exp(a - m) + np.exp(b - m))
    return np.where(np.isneginf(m), -np.inf, r)


def ctc_batch_cost_fwd_bwd(y_pred, labels, in_len, lab_len):
    """loss (B,1) and dloss/dy_pred (B,L,V) of K.ctc_batch_cost(labels, y_pred, in_len, lab_len):
    logits=log(y_pred+1e-7) -> tf.nn.ctc_loss (internal log-softmax, blank=V-1,
    ctc_merge_repeated=True, preprocess_collapse_repeated=False).  Frames t>=in_len get zero grad.
    labels may be float (utils.py:107 feeds float32)."""
    y_pred = np.asarray(y_pred, F64)
    bsz, L, V = y_pred.shape
    blank = V - 1
    loss = np.zeros((bsz, 1)); grad = np.zeros_like(y_pred)
    for b in range(bsz):
        T = int(np.ravel(in_len)[b]); U = int(np.ravel(lab_len)[b])
        lab = np.asarray(labels[b][:U]).astype(np.int64)
        ext = np.full(2 * U + 1, blank, np.int64); ext[1::2] = lab
        S = 2 * U + 1
        pe = y_pred[b, :T] + K_EPS                           # (T,V)
        den = pe.sum(-1, keepdims=True)
        lp = np.log(pe) - np.log(den)                        # log-softmax of log(p+eps)
        la = np.full((T, S), -np.inf); lb = np.full((T, S), -np.inf)
        la[0, 0] = lp[0, ext[0]]
        if S > 1:
            la[0, 1] = lp[0, ext[1]]
        can_skip = np.zeros(S, bool)
        can_skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        for t in range(1, T):
            a = la[t - 1].copy()
            a[1:] = _logsumexp(a[1:], la[t - 1, :-1])
            sk = np.full(S, -np.inf); sk[2:] = la[t - 1, :-2]
            a = np.where(can_skip, _logsumexp(a, sk), a)
            la[t] = a + lp[t, ext]
        # beta excludes the emission at t (TF convention)
        lb[T - 1, S - 1] = 0.0
        if S > 1:
            lb[T - 1, S - 2] = 0.0
        for t in range(T - 2, -1, -1):
            e = lb[t + 1] + lp[t + 1, ext]
            bb = e.copy()
            bb[:-1] = _logsumexp(bb[:-1], e[1:])
            sk = np.full(S, -np.inf); sk[:-2] = np.where(can_skip[2:], e[2:], -np.inf)
            lb[t] = _logsumexp(bb, sk)
        ll = _logsumexp(la[T - 1, S - 1], la[T - 1, S - 2]) if S > 1 else la[T - 1, 0]
        loss[b, 0] = -ll
        if not np.isfinite(ll):
            continue                                          # infeasible: TF raises; we return inf, zero grad
        occ = np.zeros((T, V))
        w = np.exp(la + lb - ll)                              # posterior of being in state s at t
        for s in range(S):
            occ[:, ext[s]] += w[:, s]
        # d(-ll)/dp = 1/den - occ/(p+eps)
        grad[b, :T] = 1.0 / den - occ / pe
    return loss, grad


# --------------------------------------------------------------------------
# VladPooling -- VLAD.py:26-49
# --------------------------------------------------------------------------
def vlad_fwd(feat, score, centers, k_centers, mode):
    """feat (B,W,H,D), score (B,W,H,K+G), centers (K+G,D) -> (B, K*D).
    softmax w/ max-subtraction (33-35), residual sum over W,H (38-42), ghost drop (44-45),
    per-cluster l2_normalize over D (47), flatten (48)."""
    feat = np.asarray(feat, F64); score = np.asarray(score, F64); centers = np.asarray(centers, F64)
    bsz = feat.shape[0]; D = feat.shape[-1]; KG = score.shape[-1]
    f = feat.reshape(bsz, -1, D); s = score.reshape(bsz, -1, KG)
    A = softmax(s)                                            # (B,L,KG)
    res = np.einsum("blk,bld->bkd", A, f) - A.sum(1)[:, :, None] * centers[None]
    if mode == "gvlad":
        res = res[:, :k_centers]
    ss = (res ** 2).sum(-1, keepdims=True)
    inv = 1.0 / np.sqrt(np.maximum(ss, L2N_EPS))
    out = res * inv
    return out.reshape(bsz, -1), (f, A, centers, res, inv, ss, feat.shape, score.shape, k_centers, mode)


def vlad_bwd(dout, cache):
    f, A, centers, res, inv, ss, fshape, sshape, k_centers, mode = cache
    bsz, L, D = f.shape; KG = A.shape[-1]
    Kk = res.shape[1]
    do = np.asarray(dout, F64).reshape(bsz, Kk, D)
    n = res * inv
    dres_k = np.where(ss >= L2N_EPS, (do - n * (n * do).sum(-1, keepdims=True)) * inv, do * inv)
    dres = np.zeros((bsz, KG, D)); dres[:, :Kk] = dres_k
    dA = np.einsum("bkd,bld->blk", dres, f) - np.einsum("bkd,kd->bk", dres, centers)[:, None, :]
    dfeat = np.einsum("blk,bkd->bld", A, dres)
    dcent = -(A.sum(1)[:, :, None] * dres).sum(0)
    dscore = A * (dA - (dA * A).sum(-1, keepdims=True))
    return dfeat.reshape(fshape), dscore.reshape(sshape), dcent


# --------------------------------------------------------------------------
# margin-softmax heads -- losses.py:28-47 / 74-91 / 121-144
# --------------------------------------------------------------------------
def _l2n_fwd(x, axis):
    ss = (x ** 2).sum(axis, keepdims=True)
    inv = 1.0 / np.sqrt(np.maximum(ss, L2N_EPS))
    return x * inv, (x * inv, inv, ss, axis)


def _l2n_bwd(dv, cache):
    v, inv, ss, axis = cache
    return np.where(ss >= L2N_EPS, (dv - v * (v * dv).sum(axis, keepdims=True)) * inv, dv * inv)


def l2_normalize_fwd(x, axis=1):
    """K.l2_normalize(x, 1) -- model.py:162."""
    return _l2n_fwd(np.asarray(x, F64), axis)


l2_normalize_bwd = _l2n_bwd


def margin_head_fwd(x, y, W, mode, s=30.0, m=0.3):
    """SphereFace / CosFace / ArcFace .call([x, y]) -> softmax probabilities.
    mode in {'sphereface','cosface','arcface'}."""
    x = np.asarray(x, F64); y = np.asarray(y, F64); W = np.asarray(W, F64)
    xn, cx = _l2n_fwd(x, 1)
    wn, cw = _l2n_fwd(W, 0)
    c = xn @ wn
    if mode == "cosface":
        t = c - m; dt = np.ones_like(c)
    else:
        lo, hi = -1.0 + K_EPS, 1.0 - K_EPS
        cc = np.clip(c, lo, hi)
        inside = (c >= lo) & (c <= hi)
        th = np.arccos(cc)
        if mode == "arcface":
            t = np.cos(th + m); dt = np.sin(th + m) / np.sin(th)
        elif mode == "sphereface":
            t = np.cos(m * th); dt = m * np.sin(m * th) / np.sin(th)
        else:
            raise ValueError(mode)
        dt = np.where(inside, dt, 0.0)
    logits = (c * (1 - y) + t * y) * s
    p = softmax(logits)
    return p, (xn, wn, cx, cw, y, dt, p, s)


def margin_head_bwd(dp, cache):
    xn, wn, cx, cw, y, dt, p, s = cache
    dp = np.asarray(dp, F64)
    dlog = p * (dp - (dp * p).sum(-1, keepdims=True))
    dc = s * dlog * ((1 - y) + y * dt)
    dxn = dc @ wn.T; dwn = xn.T @ dc
    return _l2n_bwd(dxn, cx), _l2n_bwd(dwn, cw)


def circle_loss_fwd_bwd(y_true, y_pred, gamma=256.0, margin=0.25):
    """losses.py:157-172.  Returns per-sample loss (B,) and d loss_b / d y_pred (B,A)
    (alpha_p / alpha_n are stop_gradient'ed)."""
    y = np.asarray(y_true, F64); sp = np.asarray(y_pred, F64)
    ap = np.maximum(1 + margin - sp, 0); an = np.maximum(sp + margin, 0)
    dpos, dneg = 1 - margin, margin
    z = (y * (ap * (sp - dpos)) + (1 - y) * (an * (sp - dneg))) * gamma
    zm = z.max(-1, keepdims=True)
    lse = zm[:, 0] + np.log(np.exp(z - zm).sum(-1))
    loss = y.sum(-1) * lse - (y * z).sum(-1)
    dz = softmax(z) * y.sum(-1, keepdims=True) - y
    ds = dz * gamma * (y * ap + (1 - y) * an)
    return loss, ds


def categorical_crossentropy_fwd_bwd(y_true, p):
    """Keras categorical_crossentropy on probabilities (model.py:350,354; App. A):
    p/=sum(p); clip(p,1e-7,1-1e-7); -sum(y*log p).  Returns (B,) loss and dloss_b/dp."""
    y = np.asarray(y_true, F64); p = np.asarray(p, F64)
    S = p.sum(-1, keepdims=True)
    q = p / S
    inside = (q >= K_EPS) & (q <= 1 - K_EPS)
    qc = np.clip(q, K_EPS, 1 - K_EPS)
    loss = -(y * np.log(qc)).sum(-1)
    dq = np.where(inside, -y / qc, 0.0)
    dp = (dq - (dq * q).sum(-1, keepdims=True)) / S
    return loss, dp


def global_avg_pool_fwd(x):
    """GlobalAveragePooling1D (model.py:125): mean over axis 1, no masking."""
    return np.asarray(x, F64).mean(1)


def global_avg_pool_bwd(dy, L):
    return np.repeat(dy[:, None, :], L, 1) / L


# --------------------------------------------------------------------------
# optimiser -- model.py:197 Adam(lr, decay=2e-4); model.py:163 unit_norm
# --------------------------------------------------------------------------
def adam_step(p, g, m, v, it, lr=0.01, decay=2e-4, b1=0.9, b2=0.999, eps=K_EPS):
    """Keras-2.2.4 Adam.get_updates; `it` = iterations completed before this step."""
    lr_ = lr / (1.0 + decay * it)
    t = it + 1
    lr_t = lr_ * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m2 = b1 * m + (1 - b1) * g
    v2 = b2 * v + (1 - b2) * g * g
    return p - lr_t * m2 / (np.sqrt(v2) + eps), m2, v2


def unit_norm(w):
    """keras.constraints.unit_norm(axis=0): w / (K.epsilon() + sqrt(sum_axis0 w^2))."""
    return w / (K_EPS + np.sqrt((w ** 2).sum(0, keepdims=True)))


def cal_descriptors(T, D):
    """utils.py:156-159: encoder length = ceil^5(T/2) * ceil^5(D/2)."""
    def pool(x):
        return int(np.ceil(x / 2))
    for _ in range(5):
        T, D = pool(T), pool(D)
    return T * D


def ctc_greedy_decode(y_pred, in_len):
    """K.ctc_decode(greedy=True) as used by model.py:385-389: argmax per frame, merge repeats,
    drop blank (V-1).  Returns list of int lists."""
    y_pred = np.asarray(y_pred)
    V = y_pred.shape[-1]
    out = []
    for b in range(y_pred.shape[0]):
        T = int(np.ravel(in_len)[b])
        am = y_pred[b, :T].argmax(-1)
        seq = []; prev = -1
        for a in am:
            if a != prev and a != V - 1:
                seq.append(int(a))
            prev = a
        out.append(seq)
    return out


# --------------------------------------------------------------------------- data feeder (utils.py:35-116)
# Third-party arithmetic on this row: sklearn.preprocessing.MinMaxScaler (utils.py:5-6,35-36; version unpinned by the
# reference's README).  scikit-learn IS importable in this image, so tests/test_oracle_ops.py pins feat_norm against the
# real MinMaxScaler().fit_transform -- this row's parity is pinned, unlike the TF/Keras rows.
def feat_norm(feat):
    """utils.py:35-36: MinMaxScaler(feature_range=(0,1)).fit_transform(feat) restated: per feature column over the
    utterance's frames; scale_ = 1 / (max - min) with a zero range replaced by 1; X * scale_ + (0 - min * scale_), in the
    input's float dtype and in sklearn's operation order (two roundings, no fma)."""
    feat = np.asarray(feat)
    if feat.dtype not in (np.float32, np.float64):
        feat = feat.astype(np.float64)
    if feat.shape[0] == 0:
        raise ValueError("Found array with 0 sample(s) (shape=%s) while a minimum of 1 is required by MinMaxScaler." % (feat.shape,))
    dt = feat.dtype.type
    mn = feat.min(0); rng = feat.max(0) - mn
    rng = np.where(rng == 0, dt(1), rng).astype(feat.dtype)
    scale = (dt(1) / rng).astype(feat.dtype)
    min_ = (dt(0) - mn * scale).astype(feat.dtype)
    out = feat * scale
    out = out + min_
    return out


def feat_reshape(feat, max_len=1200):
    """utils.py:39-46: truncate to max_len frames or zero-pad (float64 zeros in the reference; cast to f32 at utils.py:102)."""
    h, w = feat.shape
    if h >= max_len:
        return feat[:max_len]
    out = np.zeros((max_len, w)); out[:h] = feat
    return out


EOS_ID = 2


def text_ids_norm(ids, max_len):
    """utils.py:57-63: truncate to max_len or pad with EOS (2)."""
    ids = list(ids)[:max_len]
    return ids + [EOS_ID] * (max_len - len(ids))


def data_loader(lst, feats, ctc_enable=False, ar_enable=False, disc_enable=False, accent_dct=None, trans_dct=None,
                max_input_len=1200, max_ctc_len=72, encoder_len=100, accent_classes=8, bn=0):
    """utils.py:71-116 with the pickle loads replaced by `feats[utt]` arrays: returns (input_data, output_data)."""
    inputs = [feat_reshape(feat_norm(feats[u]), max_input_len) for u in lst]
    x = {"x_data": np.float32(np.expand_dims(np.asarray(inputs), 3))}
    y = {}
    if ctc_enable and trans_dct:
        x["x_ctc_in_len"] = np.full((len(lst), 1), encoder_len, np.int32)
        x["x_ctc_out_len"] = np.int32([[min(len(trans_dct[u]), max_ctc_len)] for u in lst])
        x["x_ctc_label"] = np.float32([text_ids_norm(trans_dct[u], max_ctc_len) for u in lst])
    if ctc_enable:
        y["y_ctc_loss"] = np.zeros([len(lst)])
    acc = None
    if ar_enable and accent_dct:
        acc = np.eye(accent_classes, dtype=np.float32)[[int(accent_dct[u]) for u in lst]]
    if ar_enable:
        y["y_accent"] = acc
    if disc_enable:
        x["x_accent"] = acc; y["y_disc"] = acc
        if bn:
            y["y_disc_bn"] = acc
    return x, y
