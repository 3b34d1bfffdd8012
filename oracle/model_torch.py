"""TEST INFRASTRUCTURE ONLY -- whole-graph CPU oracle / CPU baseline for SAR-Net.

A torch-CPU (autograd) restatement of the graph that /root/reference/model.py:204-371
builds, composed from torch primitives whose per-op results are checked against the
numpy restatement in oracle/ops_np.py (tests/test_oracle_*.py).  It provides
  * forward outputs, per-head losses, total loss (model.py:344-369 incl. the weight
    overwrite at line 361), gradients (autograd) and one Keras-Adam step, in float64
    for golden fixtures (tests/golden/) and in float32 for bench.py's cpu_baseline leg.

PARITY UNPINNED (see oracle/ops_np.py header and DESIGN.md): the reference has no
tests/fixtures and TF/Keras cannot be imported here.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from collections import OrderedDict
import math
import numpy as np
import torch
import torch.nn.functional as F

K_EPS = 1e-7
BN_EPS = 1e-3
BN_MOMENTUM = 0.99
LN_EPS = 1e-14
L2 = 1e-4


class Cfg:
    """Keyword surface of SAR_Net (model.py:204-224)."""

    def __init__(self, input_shape, ctc_enable=False, ar_enable=True, disc_enable=False, res_type="res18",
                 res_filters=64, hidden_dim=256, bn_dim=0, bpe_classes=1000, accent_classes=8, max_ctc_len=72,
                 mto=None, vlad_clusters=8, ghost_clusters=2, metric_loss="cosface", margin=0.3, lr=0.01):
        self.__dict__.update(locals()); del self.__dict__["self"]


# --------------------------------------------------------------------------- naming / init
class _Namer:
    """Keras auto-names (conv2d_N, batch_normalization_N) in construction order."""

    def __init__(self):
        self.c = {}

    def __call__(self, base):
        self.c[base] = self.c.get(base, 0) + 1
        return "%s_%d" % (base, self.c[base])


def resnet_spec(res_type, filters):
    """Layer list of resnet18_/resnet34_ (resnet.py:170-201) as a small program.
    Ops: ('conv', name, kh, kw, cin, cout, stride, padding) ('bn', name, c) ('relu',) ('maxpool',)
         ('save', slot) ('add', slot_shortcut)"""
    nm = _Namer()
    reps = {"res18": [2, 2, 2, 2], "res34": [3, 4, 6, 3]}[res_type]
    stem = filters if res_type == "res18" else 64                     # resnet.py:173 vs 191
    prog = [("conv", nm("conv2d"), 7, 7, 1, stem, 2, "same"), ("bn", nm("batch_normalization"), stem),
            ("relu",), ("maxpool",)]
    cin = stem; f = filters
    for i, r in enumerate(reps):
        for j in range(r):
            stride = 2 if (j == 0 and i != 0) else 1                   # resnet.py:96-98
            first = (i == 0 and j == 0)
            prog.append(("save", "blk_in"))
            if not first:                                             # resnet.py:111-120
                prog += [("bn", nm("batch_normalization"), cin), ("relu",)]
            prog.append(("conv", nm("conv2d"), 3, 3, cin, f, stride, "same"))
            prog += [("bn", nm("batch_normalization"), f), ("relu",),
                     ("conv", nm("conv2d"), 3, 3, f, f, 1, "same")]
            if stride > 1 or cin != f:                                 # resnet.py:75-87
                prog.append(("shortcut_conv", nm("conv2d"), 1, 1, cin, f, stride, "valid"))
            prog.append(("add",))
            cin = f
        f *= 2
    prog += [("bn", nm("batch_normalization"), cin), ("relu",)]
    return prog, cin


def _he_normal(rng, shape, fan_in):
    std = math.sqrt(2.0 / fan_in) / .87962566103423978
    w = rng.standard_normal(shape) * std
    bad = np.abs(w) > 2 * std
    while bad.any():
        w[bad] = rng.standard_normal(bad.sum()) * std
        bad = np.abs(w) > 2 * std
    return w


def _glorot(rng, shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, shape)


def _orth(rng, shape):
    rows, cols = shape[0], int(np.prod(shape[1:]))
    a = rng.standard_normal((rows, cols))
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == (rows, cols) else vt
    return q.reshape(shape)


def init_params(cfg, seed=0):
    """name -> float64 ndarray, Keras tensor layouts (Conv HWIO, Dense (in,out), GRU (in,3H)/(H,3H)/(6H)).
    Also returns the set of L2-regularised names."""
    rng = np.random.default_rng(seed)
    P = OrderedDict(); reg = set()
    prog, cout = resnet_spec(cfg.res_type, cfg.res_filters)
    for op in prog:
        if op[0] in ("conv", "shortcut_conv"):
            _, n, kh, kw, ci, co, _, _ = op
            P[n + "/kernel"] = _he_normal(rng, (kh, kw, ci, co), kh * kw * ci); reg.add(n + "/kernel")
            P[n + "/bias"] = np.zeros(co)
        elif op[0] == "bn":
            P[op[1] + "/gamma"] = np.ones(op[2]); P[op[1] + "/beta"] = np.zeros(op[2])
    H = cfg.hidden_dim

    def dense(n, i, o, bias=True, init="he", regd=True):
        P[n + "/kernel"] = _he_normal(rng, (i, o), i) if init == "he" else _glorot(rng, (i, o), i, o)
        if regd:
            reg.add(n + "/kernel")
        if bias:
            P[n + "/bias"] = np.zeros(o)
            if regd:
                reg.add(n + "/bias")

    def ln(n, c):
        P[n + "/gamma"] = np.ones(c); P[n + "/beta"] = np.zeros(c)

    def bigru(n, i):
        for d in ("forward", "backward"):
            P["%s/%s/kernel" % (n, d)] = _glorot(rng, (i, 3 * H), i, 3 * H); reg.add("%s/%s/kernel" % (n, d))
            P["%s/%s/recurrent_kernel" % (n, d)] = np.concatenate([_orth(rng, (H, H)) for _ in range(3)], 1)
            P["%s/%s/bias" % (n, d)] = np.zeros(6 * H); reg.add("%s/%s/bias" % (n, d))

    dense("CNN_LIN", cout, H); ln("CNN_LIN_LN", H)
    bigru("CRNN", H); ln("CRNN_LN", 2 * H)
    if cfg.ctc_enable:
        bigru("CTC_BIGRU", 2 * H); ln("CTC_BIGRU_LN", 2 * H)
        dense("CTC_DS", 2 * H, H); ln("CTC_DS_LN", H)
        dense("ctc_pred", H, cfg.bpe_classes)
    if cfg.ar_enable:
        dense("AR_DS", 2 * H, H); ln("AR_DS_LN", H)
        if cfg.mto == "avg":
            pooled = H
        elif cfg.mto == "bigru":
            bigru("AR_MERGE", H); pooled = 2 * H
        elif cfg.mto in ("vlad", "gvlad"):
            kg = cfg.vlad_clusters + (cfg.ghost_clusters if cfg.mto == "gvlad" else 0)
            n = cfg.mto + "_center_assignment"
            P[n + "/kernel"] = _orth(rng, (H, kg)).reshape(1, 1, H, kg) if H >= kg else _orth(rng, (kg, H)).T.reshape(1, 1, H, kg)
            P[n + "/bias"] = np.zeros(kg); reg.update([n + "/kernel", n + "/bias"])
            P[cfg.mto + "_pool/centers"] = _orth(rng, (kg, H))
            pooled = cfg.vlad_clusters * H
        else:
            raise ValueError("Please specify avg/bigru/vlad/gvlad ..")
        P["AR_BN1/gamma"] = np.ones(pooled); P["AR_BN1/beta"] = np.zeros(pooled)
        dense("AR_EMBEDDING", pooled, H)
        P["AR_BN2/gamma"] = np.ones(H); P["AR_BN2/beta"] = np.zeros(H)
        dense("AR_CF_DS1", H, 64); dense("AR_CF_DS2", 64, 64); dense("y_accent", 64, cfg.accent_classes)
        if cfg.disc_enable:
            if cfg.metric_loss == "softmax":
                dense("y_disc", H, cfg.accent_classes, bias=False)
            elif cfg.metric_loss in ("sphereface", "cosface", "arcface"):
                P["y_disc/W"] = _glorot(rng, (H, cfg.accent_classes), H, cfg.accent_classes)
            elif cfg.metric_loss == "circleloss":
                dense("y_disc", H, cfg.accent_classes, bias=False, init="glorot", regd=False)
            if cfg.bn_dim:                                                       # model.py:312-322
                dense("AR_BN_DS", H, 64)
                P["AR_BN3/gamma"] = np.ones(64); P["AR_BN3/beta"] = np.zeros(64)
                dense("bottleneck", 64, cfg.bn_dim)
                P["AR_BN4/gamma"] = np.ones(cfg.bn_dim); P["AR_BN4/beta"] = np.zeros(cfg.bn_dim)
                if cfg.metric_loss == "softmax":
                    dense("y_disc_bn", cfg.bn_dim, cfg.accent_classes, bias=False)
                elif cfg.metric_loss in ("sphereface", "cosface", "arcface"):
                    P["y_disc_bn/W"] = _glorot(rng, (cfg.bn_dim, cfg.accent_classes), cfg.bn_dim, cfg.accent_classes)
                elif cfg.metric_loss == "circleloss":
                    dense("y_disc_bn", cfg.bn_dim, cfg.accent_classes, bias=False, init="glorot", regd=False)
    return P, reg


def init_state(P):
    """BatchNormalization moving statistics (Keras: moving_mean 0, moving_variance 1) for every BN in P."""
    st = OrderedDict()
    for n in P:
        if n.endswith("/gamma") and ("batch_normalization" in n or n.startswith("AR_BN")) and not n.startswith("AR_BN_DS"):
            base = n[:-6]
            st[base + "/moving_mean"] = np.zeros(P[n].shape[0]); st[base + "/moving_variance"] = np.ones(P[n].shape[0])
    return st


# --------------------------------------------------------------------------- torch ops
def _same(n, k, s):
    out = -(-n // s); tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def conv2d(x, w, b, stride, padding):
    """x NHWC, w HWIO."""
    kh, kw = w.shape[0], w.shape[1]
    xc = x.permute(0, 3, 1, 2)
    if padding == "same":
        pt, pb = _same(x.shape[1], kh, stride); pl, pr = _same(x.shape[2], kw, stride)
        xc = F.pad(xc, (pl, pr, pt, pb))
    y = F.conv2d(xc, w.permute(3, 2, 0, 1), b, stride=stride)
    return y.permute(0, 2, 3, 1)


def maxpool(x):
    pt, pb = _same(x.shape[1], 3, 2); pl, pr = _same(x.shape[2], 3, 2)
    xc = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float("-inf"))
    return F.max_pool2d(xc, 3, 2).permute(0, 2, 3, 1)


def bn_train(x, g, b):
    ax = tuple(range(x.dim() - 1))
    mu = x.mean(ax); var = x.var(ax, unbiased=False)
    return (x - mu) / torch.sqrt(var + BN_EPS) * g + b


def bn(x, P, name, state=None, training=True):
    """BatchNormalization().  training: batch statistics; when `state` is given its moving averages are
    updated with momentum 0.99 (4-D fused path: Bessel-corrected variance; 2-D path: Keras' n / (n - (1 + eps)) sample variance).  not training: the moving
    averages normalise (Keras learning phase 0)."""
    g, b = P[name + "/gamma"], P[name + "/beta"]
    if not training:
        mm = torch.as_tensor(state[name + "/moving_mean"]).to(x.dtype); mv = torch.as_tensor(state[name + "/moving_variance"]).to(x.dtype)
        return (x - mm) / torch.sqrt(mv + BN_EPS) * g + b
    if state is not None:
        with torch.no_grad():
            ax = tuple(range(x.dim() - 1))
            n = x.numel() // x.shape[-1]
            mu = x.mean(ax); var = x.var(ax, unbiased=False)
            if x.dim() == 4 and n > 1:
                var = var * n / (n - 1)
            elif n > 1:                            # Keras 2.2.x BatchNormalization.call: variance *= n / (n - (1 + epsilon))
                var = var * n / (n - (1.0 + BN_EPS))
            for key, val in ((name + "/moving_mean", mu), (name + "/moving_variance", var)):
                old = torch.as_tensor(state[key]).to(x.dtype)
                state[key] = (BN_MOMENTUM * old + (1 - BN_MOMENTUM) * val).detach().cpu().numpy()
    return bn_train(x, g, b)


def ln(x, g, b):
    mu = x.mean(-1, keepdim=True); var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * g + b


def seq_mm(x, w):
    """The per-frame Dense / GRU-projection products (B*L rows): one name so that tools/grad_recipe_emulator.py can swap in a product whose
    BACKWARD rounds its operands; here it is the plain product."""
    return x @ w


def gru_dir(x, k, rk, bias, reverse):
    B, L, _ = x.shape; H = rk.shape[0]
    xa = seq_mm(x, k) + bias[:3 * H]
    h = x.new_zeros(B, H); outs = [None] * L
    for t in (range(L - 1, -1, -1) if reverse else range(L)):
        ha = h @ rk + bias[3 * H:]
        z = torch.sigmoid(xa[:, t, :H] + ha[:, :H]); r = torch.sigmoid(xa[:, t, H:2 * H] + ha[:, H:2 * H])
        hh = torch.tanh(xa[:, t, 2 * H:] + r * ha[:, 2 * H:])
        h = z * h + (1 - z) * hh
        outs[t] = h
    return torch.stack(outs, 1)


def bigru(x, P, n, seq=True):
    of = gru_dir(x, P[n + "/forward/kernel"], P[n + "/forward/recurrent_kernel"], P[n + "/forward/bias"], False)
    ob = gru_dir(x, P[n + "/backward/kernel"], P[n + "/backward/recurrent_kernel"], P[n + "/backward/bias"], True)
    if seq:
        return torch.cat([of, ob], -1)
    return torch.cat([of[:, -1], ob[:, 0]], -1)


def l2n(x, axis):
    return x * torch.rsqrt(torch.clamp((x * x).sum(axis, keepdim=True), min=1e-12))


def cce(y, p):
    q = p / p.sum(-1, keepdim=True)
    q = torch.clamp(q, K_EPS, 1 - K_EPS)
    return -(y * torch.log(q)).sum(-1)


def circle_loss(y, s, gamma, margin):
    ap = torch.relu(1 + margin - s.detach()); an = torch.relu(s.detach() + margin)
    z = (y * (ap * (s - (1 - margin))) + (1 - y) * (an * (s - margin))) * gamma
    return -(y * F.log_softmax(z, -1)).sum(-1)


def ctc_batch_cost(labels, y_pred, in_len, lab_len):
    V = y_pred.shape[-1]
    lp = F.log_softmax(torch.log(y_pred.transpose(0, 1) + K_EPS), -1)
    il = torch.as_tensor(np.ravel(in_len), dtype=torch.long); ll = torch.as_tensor(np.ravel(lab_len), dtype=torch.long)
    lab = torch.as_tensor(np.asarray(labels)).long()
    return F.ctc_loss(lp, lab, il, ll, blank=V - 1, reduction="none", zero_infinity=False).unsqueeze(1)


# --------------------------------------------------------------------------- graph
def forward(P, batch, cfg, collect=None, state=None, training=True):
    """P: name->torch tensor.  batch: dict as built by utils.py:102-116.  Returns dict of outputs:
    y_accent, y_disc, y_ctc_loss (as in model.py:327-338) plus 'ar' embedding, 'ctc_pred', 'cnn'."""
    dt = next(iter(P.values())).dtype
    x = torch.as_tensor(batch["x_data"]).to(dt)
    prog, _ = resnet_spec(cfg.res_type, cfg.res_filters)
    saved = None
    for op in prog:
        if op[0] == "conv":
            x = conv2d(x, P[op[1] + "/kernel"], P[op[1] + "/bias"], op[6], op[7])
        elif op[0] == "shortcut_conv":
            saved = conv2d(saved, P[op[1] + "/kernel"], P[op[1] + "/bias"], op[6], op[7])
        elif op[0] == "bn":
            x = bn(x, P, op[1], state, training)
        elif op[0] == "relu":
            x = torch.relu(x)
        elif op[0] == "maxpool":
            x = maxpool(x)
        elif op[0] == "save":
            saved = x
        elif op[0] == "add":
            x = saved + x
    out = {"cnn": x}
    B = x.shape[0]
    x = x.reshape(B, -1, x.shape[-1])                                            # CNN2SEQ model.py:252
    x = torch.tanh(seq_mm(x, P["CNN_LIN/kernel"]) + P["CNN_LIN/bias"])
    x = ln(x, P["CNN_LIN_LN/gamma"], P["CNN_LIN_LN/beta"])
    crnn = ln(bigru(x, P, "CRNN"), P["CRNN_LN/gamma"], P["CRNN_LN/beta"])
    out["crnn"] = crnn
    if cfg.ctc_enable:
        a = ln(bigru(crnn, P, "CTC_BIGRU"), P["CTC_BIGRU_LN/gamma"], P["CTC_BIGRU_LN/beta"])
        a = ln(torch.tanh(seq_mm(a, P["CTC_DS/kernel"]) + P["CTC_DS/bias"]), P["CTC_DS_LN/gamma"], P["CTC_DS_LN/beta"])
        pred = torch.softmax(seq_mm(a, P["ctc_pred/kernel"]) + P["ctc_pred/bias"], -1)
        out["ctc_pred"] = pred
        out["y_ctc_loss"] = ctc_batch_cost(batch["x_ctc_label"], pred, batch["x_ctc_in_len"], batch["x_ctc_out_len"])
    if cfg.ar_enable:
        ar = ln(torch.tanh(seq_mm(crnn, P["AR_DS/kernel"]) + P["AR_DS/bias"]), P["AR_DS_LN/gamma"], P["AR_DS_LN/beta"])
        if cfg.mto == "avg":
            ar = ar.mean(1)
        elif cfg.mto == "bigru":
            ar = bigru(ar, P, "AR_MERGE", seq=False)
        else:
            n = cfg.mto + "_center_assignment"
            score = ar @ P[n + "/kernel"][0, 0] + P[n + "/bias"]
            A = torch.softmax(score, -1)
            c = P[cfg.mto + "_pool/centers"]
            res = torch.einsum("blk,bld->bkd", A, ar) - A.sum(1)[:, :, None] * c[None]
            res = res[:, :cfg.vlad_clusters]
            ar = l2n(res, -1).reshape(B, -1)
        out["pooled"] = ar
        ar = bn(ar, P, "AR_BN1", state, training)
        ar = ar @ P["AR_EMBEDDING/kernel"] + P["AR_EMBEDDING/bias"]
        ar = bn(ar, P, "AR_BN2", state, training)
        out["ar"] = ar
        a1 = torch.relu(ar @ P["AR_CF_DS1/kernel"] + P["AR_CF_DS1/bias"])
        a1 = torch.relu(a1 @ P["AR_CF_DS2/kernel"] + P["AR_CF_DS2/bias"])
        out["y_accent"] = torch.softmax(a1 @ P["y_accent/kernel"] + P["y_accent/bias"], -1)
        if cfg.disc_enable:
            y = torch.as_tensor(batch["x_accent"]).to(dt)
            ml = cfg.metric_loss

            def disc_loss(v, name):                                              # model.py:140-166
                if ml == "softmax":
                    return torch.softmax(v @ P[name + "/kernel"], -1)
                if ml == "circleloss":
                    return l2n(v, 1) @ P[name + "/kernel"]
                c = l2n(v, 1) @ l2n(P[name + "/W"], 0)
                if ml == "cosface":
                    t = c - cfg.margin
                else:
                    th = torch.acos(torch.clamp(c, -1 + K_EPS, 1 - K_EPS))
                    t = torch.cos(th + cfg.margin) if ml == "arcface" else torch.cos(cfg.margin * th)
                return torch.softmax((c * (1 - y) + t * y) * 30.0, -1)

            out["y_disc"] = disc_loss(ar, "y_disc")
            if cfg.bn_dim:                                                       # model.py:312-322
                v = torch.relu(ar @ P["AR_BN_DS/kernel"] + P["AR_BN_DS/bias"])
                v = bn(v, P, "AR_BN3", state, training)
                v = v @ P["bottleneck/kernel"] + P["bottleneck/bias"]
                v = bn(v, P, "AR_BN4", state, training)
                out["bottleneck"] = v
                out["y_disc_bn"] = disc_loss(v, "y_disc_bn")
    return out


def loss_weights(cfg):
    """model.py:344-369 verbatim semantics, including the overwrite at line 361."""
    alpha, beta = 0.4, 0.01
    w = {}
    if cfg.ar_enable:
        w["y_accent"] = beta if cfg.disc_enable else 1.0
        if cfg.disc_enable:
            w["y_disc"] = 1 - alpha if cfg.ctc_enable else 1.0
    if cfg.ctc_enable:
        w["y_ctc_loss"] = 1 - alpha if not cfg.disc_enable else beta
    if cfg.bn_dim:
        w["y_disc_bn"] = 0.1                                                     # model.py:363-366
    return w


def losses(P, reg, out, batch, cfg):
    dt = next(iter(P.values())).dtype
    lw = loss_weights(cfg)
    parts = {}
    if "y_accent" in lw:
        parts["y_accent"] = cce(torch.as_tensor(batch["y_accent"]).to(dt), out["y_accent"]).mean()
    if "y_disc" in lw:
        y = torch.as_tensor(batch["y_disc"]).to(dt)
        if cfg.metric_loss == "circleloss":
            parts["y_disc"] = circle_loss(y, out["y_disc"], 256.0, cfg.margin).mean()
        else:
            parts["y_disc"] = cce(y, out["y_disc"]).mean()
    if "y_ctc_loss" in lw:
        parts["y_ctc_loss"] = out["y_ctc_loss"].mean()
    if "y_disc_bn" in lw:
        # model.py:364 tests `metrics != 'circleloss'` (a dict against a string): always true, so this output always gets
        # categorical_crossentropy -- with circleloss that is CE on the re-normalised raw cosine logits.  Kept verbatim.
        parts["y_disc_bn"] = cce(torch.as_tensor(batch["y_disc_bn"]).to(dt), out["y_disc_bn"]).mean()
    parts["l2"] = sum(L2 * (P[n] ** 2).sum() for n in P if n in reg)
    total = sum(lw[k] * parts[k] for k in lw) + parts["l2"]
    return total, parts


def adam_update(P, G, M, V, it, lr, decay=2e-4, b1=0.9, b2=0.999, eps=K_EPS, constrained=()):
    lr_ = lr / (1.0 + decay * it); t = it + 1
    lr_t = lr_ * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    for n in P:
        g = G[n]
        M[n] = b1 * M[n] + (1 - b1) * g
        V[n] = b2 * V[n] + (1 - b2) * g * g
        p = P[n] - lr_t * M[n] / (torch.sqrt(V[n]) + eps)
        if n in constrained:
            p = p / (K_EPS + torch.sqrt((p * p).sum(0, keepdim=True)))
        P[n] = p
    return P, M, V


class TorchSARNet:
    """Small stateful wrapper: params (torch, requires_grad), Adam state, train_on_batch()."""

    def __init__(self, cfg, params_np=None, seed=0, dtype=torch.float64):
        self.cfg = cfg
        P, self.reg = init_params(cfg, seed)
        if params_np is not None:
            P = OrderedDict((k, np.asarray(params_np[k])) for k in P)
        self.P = OrderedDict((k, torch.tensor(v, dtype=dtype)) for k, v in P.items())
        self.M = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.V = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.it = 0
        self.constrained = ("y_disc/kernel", "y_disc_bn/kernel") if (cfg.disc_enable and cfg.metric_loss == "circleloss") else ()
        self.state = init_state(P)

    def forward_backward(self, batch):
        P = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in self.P.items())
        out = forward(P, batch, self.cfg, state=self.state, training=True)
        total, parts = losses(P, self.reg, out, batch, self.cfg)
        total.backward()
        G = {k: (P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])) for k in P}
        return out, total.detach(), {k: v.detach() for k, v in parts.items()}, G

    def predict(self, batch):
        """Inference-mode forward (moving averages), no state update."""
        with torch.no_grad():
            return forward(self.P, batch, self.cfg, state=self.state, training=False)

    def evaluate(self, batch):
        with torch.no_grad():
            out = forward(self.P, batch, self.cfg, state=self.state, training=False)
            total, parts = losses(self.P, self.reg, out, batch, self.cfg)
        return out, total, parts

    def train_on_batch(self, batch):
        out, total, parts, G = self.forward_backward(batch)
        self.P, self.M, self.V = adam_update(self.P, G, self.M, self.V, self.it, self.cfg.lr,
                                             constrained=self.constrained)
        self.it += 1
        return out, total, parts, G


def synthetic_batch(cfg, B, seed=1234, full_len=False):
    """Synthetic batch per SURVEY 8d: U[0,1) fbank with zeroed tails, uniform accents,
    CTC labels in [0, V-2], label_len ~ U{4..16} clipped to what L admits."""
    rng = np.random.default_rng(seed)
    T, D, _ = cfg.input_shape
    x = rng.random((B, T, D, 1)).astype(np.float32)
    if not full_len:
        for b in range(B):
            n = rng.integers(int(math.ceil(0.6 * T)), T + 1)
            x[b, n:] = 0
    acc = rng.integers(0, cfg.accent_classes, B)
    onehot = np.eye(cfg.accent_classes, dtype=np.float32)[acc]
    batch = {"x_data": x, "y_accent": onehot}
    if cfg.disc_enable:
        batch["x_accent"] = onehot; batch["y_disc"] = onehot
        if cfg.bn_dim:
            batch["y_disc_bn"] = onehot
    if cfg.ctc_enable:
        def pool(v):
            for _ in range(5):
                v = -(-v // 2)
            return v
        L = pool(T) * pool(D)
        lab = np.full((B, cfg.max_ctc_len), 2, np.float32); ll = np.zeros((B, 1), np.int32)
        for b in range(B):
            u = int(rng.integers(min(4, max(1, L // 3)), max(2, min(16, L // 2)) + 1))
            u = min(u, cfg.max_ctc_len)
            lab[b, :u] = rng.integers(0, cfg.bpe_classes - 1, u); ll[b] = u
        batch["x_ctc_label"] = lab; batch["x_ctc_out_len"] = ll
        batch["x_ctc_in_len"] = np.full((B, 1), L, np.int32); batch["y_ctc_loss"] = np.zeros(B, np.float32)
    return batch
