"""Eager layer objects for the SAR-Net hot path.  Each mirrors a Keras layer the reference instantiates
(/root/reference/model.py:23-53) and owns an explicit backward: forward/backward are sequences of C-ABI kernel
launches on the current HIP stream (no torch autograd, no torch math).  Weight names/layouts follow Keras so that
by-name checkpoints map 1:1."""
import os

import numpy as np
import torch

from . import _lib
from .engine import ACT, current_store, gemm, pick_splitk, he_normal, glorot_uniform, orthogonal

_call = _lib.call
BN_EPS, BN_MOMENTUM, LN_EPS = 1e-3, 0.99, 1e-14


def _aligned16(*tensors):
    """every operand starts on a 16-byte boundary (what the vector loads / stores of sar_gemm_pair_f32 need)"""
    return all(t.data_ptr() % 16 == 0 for t in tensors)


class Layer:
    def __init__(self, name):
        self.name = name
        self.store = current_store()
        self.store.register_layer(self)

    def __init_subclass__(cls, **kw):
        """Every layer remembers the tensor its last forward() produced: `layer.output` is what keras' `layer.output`
        evaluates to for the current batch (model.py:380-383 sub_model builds on it)."""
        super().__init_subclass__(**kw)
        fwd = cls.__dict__.get("forward")
        if fwd is not None:
            def forward(self, *a, _fwd=fwd, **k):
                out = _fwd(self, *a, **k)
                self.output = out
                return out
            forward.__doc__ = fwd.__doc__
            cls.forward = forward

    def buf(self, key, shape, dtype=torch.float32):
        return self.store.bufs.get(self.name + "/" + key, shape, dtype)

    def scratch(self, key, shape, dtype=torch.float32):
        return self.store.bufs.get("scratch/" + key, shape, dtype)

    def count_params(self):
        return sum(r.size for r in self.params())

    def params(self):
        return [r for r in self.store.refs if r.name.startswith(self.name + "/")]


class Dense(Layer):
    """keras Dense via DS() (model.py:35-42): y = act(x W + b), he_normal kernel, l2(1e-4) on kernel and bias."""

    def __init__(self, in_dim, units, activation=None, use_bias=True, name=None, l2=1e-4, init="he_normal",
                 constraint=None):
        super().__init__(name)
        rng = self.store.rng
        w = he_normal(rng, (in_dim, units), in_dim) if init == "he_normal" else glorot_uniform(rng, (in_dim, units), in_dim, units)
        self.kernel = self.store.add(name + "/kernel", (in_dim, units), w, l2, constraint)
        self.bias = self.store.add(name + "/bias", (units,), np.zeros(units), l2) if use_bias else None
        self.in_dim, self.units, self.activation = in_dim, units, activation
        self.output_shape = (None, units)

    def forward(self, x):
        R = x.numel() // self.in_dim
        self.x, self.R = x, R
        y = self.buf("y", (R, self.units))
        b = self.bias.w if self.bias is not None else None
        if self.activation == "softmax":
            z = self.buf("z", (R, self.units))
            gemm(x, self.kernel.w, z, b, R, self.units, self.in_dim, self.in_dim, self.units, self.units, 1, 0)
            _call("sar_softmax_fwd_f32", z, y, R, self.units)
        else:
            sk = pick_splitk(R, self.units, self.in_dim) if ACT[self.activation] == 0 else 1
            if sk > 1 and not (self.store.training and self.store.zero_at_step_start(y)):
                _call("sar_fill_f32", y, 0.0, y.numel())
            gemm(x, self.kernel.w, y, b, R, self.units, self.in_dim, self.in_dim, self.units, self.units, 1, 0,
                 ACT[self.activation], 0, sk)
        self.y = y
        return y

    def backward(self, dy, need_dx=True, accumulate_dx=False, dx_out=None, dz_ready=False):
        R = self.R
        defer = self.store.deferring_wgrads() and self.in_dim >= 64 and self.units >= 64 and R >= 64 and (self.in_dim | self.units | R) % 4 == 0
        # a deferred weight gradient reads dz when the group is launched: the layer's own buffer then, not the shared scratch
        dzbuf = (lambda: self.buf("dz", (R, self.units))) if defer else (lambda: self.scratch("dz%dx%d" % (R, self.units), (R, self.units)))
        if dz_ready:                                               # the caller (LayerNorm.backward(act=...)) already applied act'
            dz = dy
        elif self.activation == "softmax":
            dz = dzbuf()
            _call("sar_softmax_bwd_f32", dy, self.y, dz, R, self.units)
        elif ACT[self.activation] != 0:
            dz = dzbuf()
            _call("sar_act_bwd_f32", dy, self.y, dz, R * self.units, ACT[self.activation])
        else:
            dz = dy
        sk = pick_splitk(self.in_dim, self.units, R)
        wg = (self.x, dz, self.kernel.g, self.bias.g if self.bias is not None else None, self.in_dim, self.units, R, self.in_dim,
              self.units, self.units, sk)
        if defer:
            self.store.defer_wgrad(*wg)
        else:
            _call("sar_gemm_wgrad_f32", *wg)
        if not need_dx:
            return None
        dx = dx_out if dx_out is not None else self.buf("dx", (R, self.in_dim))
        gs = self.store.bwd_gemm_scale(R)
        # the split-precision mode's backward: one pass on fp16 operands (csrc/gemm.hip: F16).  (16-byte aligned operands only: the entry
        # returns SAR_EINVAL for a view that is not, and the per-product path below takes any -- ADVICE r5)
        if gs > 0 and (R | self.in_dim | self.units) % 64 == 0 and _aligned16(dz, self.kernel.w, dx):
            _call("sar_gemm_pair_f32", dz, self.kernel.w, None, None, dx, R, self.in_dim, self.units, self.units, self.units, self.in_dim, 1,
                  1 if accumulate_dx else 0, gs, 1.0)
        else:
            gemm(dz, self.kernel.w, dx, None, R, self.in_dim, self.units, self.units, self.units, self.in_dim, 1, 1, 0,
                 1 if accumulate_dx else 0)
        return dx


class LayerNorm(Layer):
    """keras_layer_normalization.LayerNormalization() (model.py:32-33): last-axis, eps = 1e-14."""

    def __init__(self, dim, name=None):
        super().__init__(name)
        self.gamma = self.store.add(name + "/gamma", (dim,), np.ones(dim))
        self.beta = self.store.add(name + "/beta", (dim,), np.zeros(dim))
        self.dim = dim; self.output_shape = (None, dim)

    def forward(self, x):
        R = x.numel() // self.dim
        self.x, self.R = x, R
        y = self.buf("y", (R, self.dim)); self.mean = self.buf("mean", (R,)); self.rstd = self.buf("rstd", (R,))
        _call("sar_layernorm_fwd_f32", x, self.gamma.w, self.beta.w, y, self.mean, self.rstd, R, self.dim, LN_EPS)
        return y

    def backward(self, dy, accumulate_dx=False, dx_out=None, act=None):
        """act: activation of the Dense whose OUTPUT this layer normalises ('tanh'): its backward is applied here as well, the
        result is d/dz of that Dense (pass it on with Dense.backward(..., dz_ready=True))."""
        dx = dx_out if dx_out is not None else self.buf("dx", (self.R, self.dim))
        _call("sar_layernorm_bwd_act_f32", dy, self.x, self.gamma.w, self.mean, self.rstd, dx, self.gamma.g, self.beta.g,
              self.R, self.dim, 1 if accumulate_dx else 0, ACT[act])
        return dx


_FUSE_DENSE_LN = os.environ.get("SAR_DENSE_LN", "1") != "0"


def dense_ln_forward(dense, ln, x):
    """DS(..., 'tanh') -> LN (model.py:253-254, 265-266, 279-280): ONE launch (sar_dense_tanh_ln_f32, csrc/gemm.hip) where the fused kernel takes
    the shape -- 256 units, bias, in_dim a multiple of 64, split-precision GEMM arithmetic -- else the two launches.  Both layers end up with exactly
    the tensors their own forward() leaves for the backward (Dense: x, y = tanh's output; LayerNorm: x = that y, mean, rstd) and `.output`."""
    R = x.numel() // dense.in_dim
    if (_FUSE_DENSE_LN and dense.activation == "tanh" and dense.bias is not None and dense.units == 256 and ln.dim == 256 and dense.in_dim % 64 == 0
            and _lib.get_gemm_math() == "bf16x3" and x.is_contiguous() and _aligned16(x, dense.kernel.w, dense.bias.w, ln.gamma.w, ln.beta.w)):
        dense.x, dense.R = x, R
        a = dense.buf("y", (R, dense.units)); dense.y = a; dense.output = a
        ln.x, ln.R = a, R
        y = ln.buf("y", (R, ln.dim)); ln.mean = ln.buf("mean", (R,)); ln.rstd = ln.buf("rstd", (R,))
        _call("sar_dense_tanh_ln_f32", x, dense.kernel.w, dense.bias.w, ln.gamma.w, ln.beta.w, a, y, ln.mean, ln.rstd, R, dense.in_dim, dense.units, LN_EPS)
        ln.output = y
        return y
    return ln.forward(dense.forward(x))


class BiGRU(Layer):
    """Bidirectional(CuDNNGRU(hidden, return_sequences=seq), merge_mode='concat') (model.py:44-50)."""

    def __init__(self, in_dim, hidden, seq=True, name=None, l2=1e-4):
        super().__init__(name)
        rng = self.store.rng
        H = hidden
        self.p = {}
        for d in ("forward", "backward"):
            k = self.store.add("%s/%s/kernel" % (name, d), (in_dim, 3 * H), glorot_uniform(rng, (in_dim, 3 * H), in_dim, 3 * H), l2)
            u = self.store.add("%s/%s/recurrent_kernel" % (name, d), (H, 3 * H),
                               np.concatenate([orthogonal(rng, (H, H)) for _ in range(3)], 1))
            b = self.store.add("%s/%s/bias" % (name, d), (6 * H,), np.zeros(6 * H), l2)
            self.p[d] = (k, u, b)
        self.in_dim, self.H, self.seq = in_dim, H, seq
        self.output_shape = (None, None, 2 * H) if seq else (None, 2 * H)

    def forward(self, x, B, L):
        H, I = self.H, self.in_dim
        self.x, self.B, self.L = x, B, L
        R = B * L
        xa = [self.buf("xa_f", (R, 3 * H)), self.buf("xa_b", (R, 3 * H))]
        # both directions' input projections in one launch (two ~1 us GEMMs would queue behind each other)
        gb = _lib.GemmBatch([(x, self.p[d][0].w, xa[i], self.p[d][2].w) for i, d in enumerate(("forward", "backward"))])
        _call("sar_gemm_batched_f32", gb, 2, R, 3 * H, I, I, 3 * H, 3 * H, 1, 0, 0, 0, 1)
        self.out = self.buf("out", (B, L, 2 * H)); self.gates = self.buf("gates", (2, B, L, 4, H)); self.hprev = self.buf("hprev", (2, B, L, H))
        (kf, uf, bf), (kb, ub, bb) = self.p["forward"], self.p["backward"]
        # "bf16" mode at the reference's hidden_dim: recurrent kernels packed as bf16 MFMA fragments, resident on chip (gru_mfma.hip)
        # "bf16x3" (fp32-class) mode: the same kernels with fp16 operands (|U| <= 1, |h| < 1: 1.3e-4 on the outputs, tools/recipe_emulator.py)
        gm = _lib.get_gemm_math()
        self.mfma = H == 256 and gm in ("bf16", "bf16x3")
        self.msfx = "f16" if gm == "bf16x3" else "bf16"
        if self.mfma:
            self.pack = self.buf("u_pack", (2 * 2 * 3 * H * H,), torch.float16 if self.msfx == "f16" else torch.bfloat16)
            _call("sar_bigru_pack_" + self.msfx, uf.w, ub.w, self.pack, H)
            _call("sar_bigru_fwd_" + self.msfx, xa[0], xa[1], self.pack, bf.w[3 * H:], bb.w[3 * H:], self.out, self.gates, self.hprev, B, L, H)
        elif H > 256:
            # past the persistent kernels' 256 units: one batched GEMM (ha = h U, both directions) + one gate launch per time step (csrc/gru.hip)
            h = self.buf("h_state", (2, B, H)); ha = self.buf("ha", (2, B, 3 * H))
            _call("sar_fill_f32", h, 0.0, h.numel())
            gb = _lib.GemmBatch([(h[0], uf.w, ha[0], None), (h[1], ub.w, ha[1], None)])
            for s in range(L):
                _call("sar_gemm_batched_f32", gb, 2, B, 3 * H, H, H, 3 * H, 3 * H, 1, 0, 0, 0, 1)
                _call("sar_bigru_step_fwd_f32", xa[0], xa[1], ha, bf.w[3 * H:], bb.w[3 * H:], h, self.out, self.gates, self.hprev, B, L, H, s)
        else:
            _call("sar_bigru_fwd_f32", xa[0], xa[1], uf.w, ub.w, bf.w[3 * H:], bb.w[3 * H:], self.out, self.gates, self.hprev, B, L, H)
        if self.seq:
            return self.out
        y = self.buf("last", (B, 2 * H))
        _call("sar_bigru_last_fwd_f32", self.out, y, B, L, H)
        return y

    def backward(self, dy, need_dx=True, accumulate_dx=False, dx_out=None):
        H, I, B, L = self.H, self.in_dim, self.B, self.L
        R = B * L
        if not self.seq:
            dout = self.scratch("gru_dout%d_%d_%d" % (B, L, H), (B, L, 2 * H))
            _call("sar_bigru_last_bwd_f32", dy, dout, B, L, H)
        else:
            dout = dy
        defer = self.store.deferring_wgrads() and I % 64 == 0 and H % 64 == 0 and R >= 64
        if defer:                                                  # read when the group is launched: this layer's own buffers
            dai = self.buf("dai", (2, B, L, 3 * H)); dah = self.buf("dah", (2, B, L, 3 * H))
        else:
            dai = self.scratch("gru_dai%d_%d_%d" % (B, L, H), (2, B, L, 3 * H)); dah = self.scratch("gru_dah%d_%d_%d" % (B, L, H), (2, B, L, 3 * H))
        if self.mfma and self.msfx == "f16":
            # fp16 operands: the pre-activation gradients enter the MFMA times a power-of-two scale (the trunk's rule: 64 x batch)
            _call("sar_bigru_bwd_f16", dout, self.gates, self.hprev, self.pack, dai, dah, B, L, H, float(2 ** (int(round(np.log2(64.0 * max(B, 1)))) + self.store.gscale_shift)))
        elif self.mfma:
            _call("sar_bigru_bwd_bf16", dout, self.gates, self.hprev, self.pack, dai, dah, B, L, H)
        elif H > 256:
            dh = self.buf("dh_state", (2, B, H)); dstep = self.buf("dstep", (2, B, 3 * H))
            _call("sar_fill_f32", dh, 0.0, dh.numel())
            gb = _lib.GemmBatch([(dstep[0], self.p["forward"][1].w, dh[0], None), (dstep[1], self.p["backward"][1].w, dh[1], None)])
            for s in range(L - 1, -1, -1):
                _call("sar_bigru_step_bwd_f32", dout, self.gates, self.hprev, dh, dai, dah, dstep, B, L, H, s)
                if s > 0:                                              # dh_{s-1} += dstep U^T (U stored (H, 3H): the [N][K] operand)
                    _call("sar_gemm_batched_f32", gb, 2, B, H, 3 * H, 3 * H, 3 * H, H, 1, 1, 0, 1, 1)
        else:
            UT = [self.scratch("gru_UT0_%d" % H, (3 * H, H)), self.scratch("gru_UT1_%d" % H, (3 * H, H))]
            for i, d in enumerate(("forward", "backward")):
                _call("sar_transpose_f32", self.p[d][1].w, UT[i], H, 3 * H)
            _call("sar_bigru_bwd_f32", dout, self.gates, self.hprev, UT[0], UT[1], dai, dah, B, L, H)
        dx = None
        if need_dx:
            dx = dx_out if dx_out is not None else self.buf("dx", (R, I))
        sk = pick_splitk(I, 3 * H, R)
        sku = pick_splitk(H, 3 * H, R)
        # the weight gradients of the four kernels (input / recurrent x two directions): one launch when they have one shape
        # (in_dim == hidden: the reference's CNN_LIN -> CRNN), else one per kind
        ks = [(self.x, dai[i], self.p[d][0].g, self.p[d][2].g[:3 * H]) for i, d in enumerate(("forward", "backward"))]
        us = [(self.hprev[i], dah[i], self.p[d][1].g, self.p[d][2].g[3 * H:]) for i, d in enumerate(("forward", "backward"))]
        if defer:
            for (x_, dy_, dw_, db_), in_, sk_ in [(q, I, sk) for q in ks] + [(q, H, sku) for q in us]:
                self.store.defer_wgrad(x_, dy_, dw_, db_, in_, 3 * H, R, in_, 3 * H, 3 * H, sk_)
        elif I == H and sk == sku:
            _call("sar_gemm_wgrad_batched_f32", _lib.GemmBatch(ks + us), 4, I, 3 * H, R, I, 3 * H, 3 * H, sk)
        else:
            _call("sar_gemm_wgrad_batched_f32", _lib.GemmBatch(ks), 2, I, 3 * H, R, I, 3 * H, 3 * H, sk)
            _call("sar_gemm_wgrad_batched_f32", _lib.GemmBatch(us), 2, H, 3 * H, R, H, 3 * H, 3 * H, sku)
        if (need_dx and (R | I | H) % 64 == 0 and _lib.get_gemm_math() in ("bf16", "bf16x3") and os.environ.get("SAR_GEMM_PAIR", "1") == "1"
                and _aligned16(dai[0], dai[1], self.p["forward"][0].w, self.p["backward"][0].w, dx)):
            # both directions' products into dx in ONE launch (two operand pairs along k; csrc/gemm.hip: GemmX) -- they were two, the second a
            # read-modify-write of dx; in the split-precision mode's backward arithmetic where that applies
            _call("sar_gemm_pair_f32", dai[0], self.p["forward"][0].w, dai[1], self.p["backward"][0].w, dx, R, I, 3 * H, 3 * H, 3 * H, I, 1,
                  1 if accumulate_dx else 0, self.store.bwd_gemm_scale(), 1.0)
        elif need_dx:
            for i, d in enumerate(("forward", "backward")):
                gemm(dai[i], self.p[d][0].w, dx, None, R, I, 3 * H, 3 * H, 3 * H, I, 1, 1, 0, 1 if (accumulate_dx or i == 1) else 0)
        return dx


class BatchNorm1d(Layer):
    """BatchNormalization() on a 2-D (B, C) tensor (model.py:286,289): batch statistics when training, moving averages otherwise."""

    def __init__(self, dim, name=None):
        super().__init__(name)
        self.gamma = self.store.add(name + "/gamma", (dim,), np.ones(dim))
        self.beta = self.store.add(name + "/beta", (dim,), np.zeros(dim))
        self.moving_mean = self.store.add_state(name + "/moving_mean", (dim,), np.zeros(dim))
        self.moving_var = self.store.add_state(name + "/moving_variance", (dim,), np.ones(dim))
        self.dim = dim; self.output_shape = (None, dim)

    def forward(self, x, B):
        self.x, self.B = x, B
        y = self.buf("y", (B, self.dim)); self.mean = self.buf("mean", (self.dim,)); self.rstd = self.buf("rstd", (self.dim,))
        if not self.store.training:                                          # learning phase 0: moving averages
            _call("sar_bn_infer_f32", x, self.gamma.w, self.beta.w, self.moving_mean.w, self.moving_var.w, y, B, self.dim, BN_EPS, 0)
            return y
        _call("sar_bn1d_fwd_f32", x, self.gamma.w, self.beta.w, y, self.mean, self.rstd, self.moving_mean.w,
              self.moving_var.w, B, self.dim, BN_EPS, BN_MOMENTUM)
        return y

    def backward(self, dy):
        dx = self.buf("dx", (self.B, self.dim))
        _call("sar_bn1d_bwd_f32", dy, self.x, self.gamma.w, self.mean, self.rstd, dx, self.gamma.g, self.beta.g, self.B, self.dim)
        return dx


class GlobalAveragePooling1D(Layer):
    """model.py:125."""

    def __init__(self, dim, name=None):
        super().__init__(name)
        self.dim = dim; self.output_shape = (None, dim)

    def forward(self, x, B, L):
        self.B, self.L = B, L
        y = self.buf("y", (B, self.dim))
        _call("sar_avgpool1d_fwd_f32", x, y, B, L, self.dim)
        return y

    def backward(self, dy, accumulate_dx=False, dx_out=None):
        dx = dx_out if dx_out is not None else self.buf("dx", (self.B * self.L, self.dim))
        _call("sar_avgpool1d_bwd_f32", dy, dx, self.B, self.L, self.dim, 1 if accumulate_dx else 0)
        return dx


class L2Normalize(Layer):
    """Lambda(lambda x: K.l2_normalize(x, 1)) (model.py:162)."""

    def __init__(self, dim, name=None):
        super().__init__(name)
        self.dim = dim; self.output_shape = (None, dim)

    def forward(self, x, B):
        self.B = B
        self.v = self.buf("v", (B, self.dim)); self.inv = self.buf("inv", (B,))
        _call("sar_l2norm_rows_fwd_f32", x, self.v, self.inv, B, self.dim)
        return self.v

    def backward(self, dv, accumulate_dx=False, dx_out=None):
        dx = dx_out if dx_out is not None else self.buf("dx", (self.B, self.dim))
        _call("sar_l2norm_rows_bwd_f32", dv, self.v, self.inv, dx, self.B, self.dim, 1 if accumulate_dx else 0)
        return dx
