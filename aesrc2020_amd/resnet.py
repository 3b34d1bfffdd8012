"""Pre-activation ResNet-18/34 trunk of SAR-Net, MI355X-native (mirrors /root/reference/resnet.py).

Same public names as the reference (`resnet18_`, `resnet34_`, ... resnet.py:170-249), same topology:
7x7/2 stem -> BN -> ReLU -> MaxPool 3x3/2 'same' -> 4 stages of basic_block ([2,2,2,2] / [3,4,6,3]) with filters
F,2F,4F,8F, stride 2 at the first block of stages 1-3 (resnet.py:96-98), 1x1 'valid' conv shortcuts where the shape
changes (resnet.py:75-87; res34 hard-codes a 64-channel stem, resnet.py:191, so stage0/block0 gets one too),
final BN+ReLU, no global pooling (resnet.py:178-181).

Execution plan (what is different from the reference): BatchNormalization+ReLU in front of a conv is never
materialised -- batch statistics are reduced once (sar_bn_stats), folded to per-channel scale/shift, and applied
inside the conv's operand load (fwd and wgrad); the residual Add is the conv epilogue; the stem's BN+ReLU is fused
with the max-pool.  Backward mirrors it: dgrad -> (BN+ReLU backward reduce, apply [+ shortcut gradient]) -> wgrad.
Layers are auto-named conv2d_N / batch_normalization_N in construction order like Keras does for the unnamed
reference layers (checkpoint row).
"""
import os

import numpy as np
import torch

from . import _lib
from .engine import current_store, he_normal, same_pads
from .layers import Layer, BN_EPS, BN_MOMENTUM

_call = _lib.call


class _BN:
    """BatchNormalization(axis=3) parameters + batch-statistics state for one use site (resnet.py:22-26)."""

    def __init__(self, store, name, C):
        self.name, self.C, self.store = name, C, store
        self.gamma = store.add(name + "/gamma", (C,), np.ones(C))
        self.beta = store.add(name + "/beta", (C,), np.zeros(C))
        self.moving_mean = store.add_state(name + "/moving_mean", (C,), np.zeros(C))
        self.moving_var = store.add_state(name + "/moving_variance", (C,), np.ones(C))

    def _b(self, key, shape, dtype=torch.float32):
        return self.store.bufs.get(self.name + "/" + key, shape, dtype)

    def prepare(self):
        C = self.C
        self.mean, self.rstd, self.scale, self.shift = (self._b(k, (C,)) for k in ("mean", "rstd", "scale", "shift"))
        self.sums = self._b("sums", (_lib.bn_workspace_doubles(C),), torch.float64)

    def stats(self, x, R):
        C = self.C
        self.prepare()
        if not self.store.training:                                          # learning phase 0: fold the moving averages
            _call("sar_bn_fold_f32", self.gamma.w, self.beta.w, self.moving_mean.w, self.moving_var.w, self.scale, self.shift, C, BN_EPS)
            return
        _call("sar_bn_stats_f32", x, self.gamma.w, self.beta.w, self.sums, self.mean, self.rstd, self.scale, self.shift,
              self.moving_mean.w, self.moving_var.w, R, C, BN_EPS, BN_MOMENTUM, 1)

    # ---- bf16 trunk (csrc/conv_bf16.hip): statistics through bins of fp32 atomics, constants recomputed by every consumer
    def prepare16(self, arena, off, det=False):
        """Claim this BatchNorm's forward / backward bins in the trunk's statistics arena (zeroed once per step).  det: the forward
        bins are 64-bit fixed point (csrc/trunk16.h: fix28) -- twice the words."""
        n = _lib.bn_bins() * 2 * self.C
        nf = 2 * n if det else n
        self.det = bool(det)
        self.bins, self.bbins = arena[off:off + nf], arena[off + nf:off + nf + n]
        self.saved = self._b("saved", (4, self.C))
        return off + nf + n

    def ref(self, R, stats, write, ginv=1.0):
        """sarBnRef for a consumer kernel.  stats: 'fwd' = batch statistics from the bins (training forward), 'saved' = the
        saved mean/rstd/scale/shift (backward consumers; inference after fold16), 'bwd' = saved + backward sums.  ginv: 1 / (the
        gradient scale the backward sums carry) -- the fp16 trunk's static scale leaves the parameter gradients here."""
        r = _lib.BnRef()
        r.gamma, r.beta, r.saved = self.gamma.w.data_ptr(), self.beta.w.data_ptr(), self.saved.data_ptr()
        r.bins = self.bins.data_ptr() if stats == "fwd" else None
        r.bbins = self.bbins.data_ptr() if stats == "bwd" else None
        r.moving_mean, r.moving_var = self.moving_mean.w.data_ptr(), self.moving_var.w.data_ptr()
        r.dgamma, r.dbeta = self.gamma.g.data_ptr(), self.beta.g.data_ptr()
        r.inv_n, r.unbias, r.eps, r.momentum = 1.0 / R, R / max(R - 1.0, 1.0), BN_EPS, BN_MOMENTUM
        r.write = int(write); r.ginv = float(ginv)
        r.det = int(getattr(self, "det", False) and stats == "fwd")
        return r

    def fold16(self):
        """Inference: moving averages -> saved scale/shift (rows 2, 3); consumers then run with stats='saved'."""
        _call("sar_bn_fold_f32", self.gamma.w, self.beta.w, self.moving_mean.w, self.moving_var.w, self.saved[2], self.saved[3], self.C, BN_EPS)

    def bwd(self, dact, x, add, dx, R, relu=1):
        """dx = d/dx [relu](bn(x)) given dact; (+ add).  Parameter gradients accumulate."""
        _call("sar_bn_bwd_f32", dact, x, self.scale, self.shift, self.mean, self.rstd, self.sums, add, dx,
              self.gamma.g, self.beta.g, R, self.C, relu)


class _Conv:
    """Conv2D(he_normal, l2(1e-4) on the kernel only, bias present) (resnet.py:39-42, 60-63, 82-87, 113-117)."""

    def __init__(self, store, name, k, cin, cout, stride, padding):
        self.name, self.k, self.cin, self.cout, self.stride, self.padding = name, k, cin, cout, stride, padding
        self.kernel = store.add(name + "/kernel", (k, k, cin, cout), he_normal(store.rng, (k, k, cin, cout), k * k * cin), 1e-4)
        self.bias = store.add(name + "/bias", (cout,), np.zeros(cout))
        self.store = store
        # bf16 (hi, lo) planes of the kernel for the split-precision kernels (ResNetTrunk.split_weights fills them once per step)
        self.wp_ok = cin % 32 == 0 and cout % 64 == 0
        self.wt = self.wd = (None, None)
        self.use_planes = False
        # 2-byte trunk (ResNetTrunk sets them): entry-point suffix "bf16" | "f16", torch dtype, 1 / gradient scale
        self.sfx, self.t16, self.ginv = "bf16", torch.bfloat16, 1.0
        self.bsfx = "bf16"                                                   # entry-point suffix of the INPUT-GRADIENT launches ("f16a": the mixed backward of the split-precision mode)
        self.epi_fwd = 1                                                     # 3: deterministic (fixed-point) statistics out of the epilogue

    def planes(self):
        K = self.k * self.k * self.cin
        b = self.store.bufs
        self.wt = tuple(b.get("%s/wt_%s" % (self.name, h), (self.cout, K), torch.bfloat16) for h in ("hi", "lo"))
        self.wd = tuple(b.get("%s/wd_%s" % (self.name, h), (K, self.cout), torch.bfloat16) for h in ("hi", "lo"))
        return [self.kernel.w.data_ptr()] + [t.data_ptr() for t in self.wt + self.wd] + [K, self.cout]

    # ---- bf16 trunk
    def streams16(self):
        """Packed bf16 weight streams (forward, input gradient) + this conv's row of the pack table."""
        taps = self.k * self.k
        npl = 2 if self.sfx == "x3" else 1                                  # split precision: hi and lo fragment of every 16-channel block
        npd = 1 if self.bsfx == "f16a" else npl                             # mixed backward: the input-gradient stream is one plane of fp16 fragments
        nf = (self.cout // 32) * (npl * taps * self.cin // 16 + 8) * 512    # 2-byte elements
        nd = (self.cin // 32) * (npd * taps * self.cout // 16 + 8) * 512
        b = self.store.bufs
        self.wf = b.get(self.name + "/wf16", (nf,), self.t16)
        self.wd = b.get(self.name + "/wd16", (nd,), torch.float16 if self.bsfx == "f16a" else self.t16)
        return [self.kernel.w.data_ptr(), self.wf.data_ptr(), self.wd.data_ptr(), taps, self.cin, self.cout], -(-(nf + nd) // 8 // 256)

    def _x3_tail(self, side2, side_scale, sfx=None):
        """sar_conv2d_x3's two extra arguments: the fill's result as an fp32 tensor (side2), the scale of its fp16 copy (side)."""
        return (side2, float(side_scale)) if (sfx or self.sfx) == "x3" else ()

    def fwd16_args(self, x, y, B, fill=0, fbn=None, res=None, e_bins=None, side=None, side2=None, side_scale=1.0):
        return (x, None, None, self.wf, self.bias.w, res, y, side, B, self.H, self.W, self.cin, self.Ho, self.Wo, self.cout,
                self.k, self.stride, 1, self.pt, self.pl, fill, fbn, self.epi_fwd if e_bins is not None else 0, e_bins, None, None) + \
            self._x3_tail(side2, side_scale)

    def fwd16(self, x, y, B, **kw):
        _call("sar_conv2d_" + self.sfx, *self.fwd16_args(x, y, B, **kw))

    def dgrad16(self, dy, dy2, dx, B, fill=0, fbn=None, res=None, e_bins=None, e_y=None, e_saved=None, side=None, dy3=None, side2=None, side_scale=1.0):
        """dx = conv_transpose(T(dy)) (+ res); T = BatchNorm backward of (dy, dy2) through fbn when fill == 2, + dy3 when fill == 3
        (written back to `side` for the weight gradient and the other consumers of that gradient); the epilogue leaves the backward
        sums of the BatchNorm in FRONT of this conv (input e_y) in e_bins."""
        _call("sar_conv2d_" + self.bsfx, *self.dgrad16_args(dy, dy2, dx, B, fill, fbn, res, e_bins, e_y, e_saved, side, dy3, side2, side_scale))

    def dgrad16_args(self, dy, dy2, dx, B, fill=0, fbn=None, res=None, e_bins=None, e_y=None, e_saved=None, side=None, dy3=None, side2=None,
                     side_scale=1.0):
        return (dy, dy2, dy3, self.wd, None, res, dx, side, B, self.Ho, self.Wo, self.cout, self.H, self.W, self.cin,
                self.k, 1, self.stride, self.k - 1 - self.pt, self.k - 1 - self.pl, fill, fbn, 2 if e_bins is not None else 0,
                e_bins, e_y, e_saved) + self._x3_tail(side2, side_scale, self.bsfx)

    def wgrad16(self, x, dy, B, bias_grad):
        """Descriptor of this conv's weight gradient on plain operands (include/sarnet.h: sarWgradConv); the trunk batches them
        (ResNetTrunk._wgrad16)."""
        return dict(x=x, dy=dy, dw=self.kernel.g, dbias=self.bias.g if bias_grad else None, B=B, H=self.H, W=self.W, Cin=self.cin,
                    Ho=self.Ho, Wo=self.Wo, Cout=self.cout, K=self.k, s=self.stride, pt=self.pt, pl=self.pl, ginv=self.ginv)

    def out_hw(self, H, W):
        if self.padding == "same":
            (ho, self.pt), (wo, self.pl) = same_pads(H, self.k, self.stride), same_pads(W, self.k, self.stride)
        else:
            ho, wo = (H - self.k) // self.stride + 1, (W - self.k) // self.stride + 1
            self.pt = self.pl = 0
        self.H, self.W, self.Ho, self.Wo = H, W, ho, wo
        return ho, wo

    def fwd(self, x, y, B, bn=None, residual=None, bn_out=None):
        """y = conv(relu(bn(x))) + bias (+ residual).  bn_out: the BatchNorm that consumes y next -- its batch statistics are
        produced by the same call (conv epilogue where the launch has no split-K), training mode only."""
        args = (x, self.kernel.w, self.bias.w, bn.scale if bn else None, bn.shift if bn else None, residual, y, B, self.H, self.W,
                self.cin, self.Ho, self.Wo, self.cout, self.k, self.k, self.stride, self.pt, self.pl)
        wt = self.wt if self.use_planes else (None, None)
        if bn_out is None:
            _call("sar_conv2d_fwd_ex_f32", *args, wt[0], wt[1], None, None, None, None, None, None, None, None, None, BN_EPS, BN_MOMENTUM, 1)
        else:
            bn_out.prepare()
            _call("sar_conv2d_fwd_ex_f32", *args, wt[0], wt[1], bn_out.gamma.w, bn_out.beta.w, bn_out.sums, bn_out.mean, bn_out.rstd,
                  bn_out.scale, bn_out.shift, bn_out.moving_mean.w, bn_out.moving_var.w, BN_EPS, BN_MOMENTUM, 1)

    def dgrad(self, dy, dx, B, accumulate=0):
        wd = self.wd if self.use_planes else (None, None)
        _call("sar_conv2d_dgrad_ex_f32", dy, self.kernel.w, dx, B, self.H, self.W, self.cin, self.Ho, self.Wo, self.cout,
              self.k, self.k, self.stride, self.pt, self.pl, accumulate, wd[0], wd[1])

    def wgrad(self, x, dy, B, bn=None, bias_grad=True):
        """Kernel gradient (+ bias gradient).  A conv whose output feeds a BatchNorm directly has an exactly-zero bias
        gradient (BN backward outputs sum to zero per channel), so callers pass bias_grad=False there."""
        _call("sar_conv2d_wgrad_f32", x, dy, bn.scale if bn else None, bn.shift if bn else None, self.kernel.g,
              self.bias.g if bias_grad else None, B, self.H, self.W, self.cin, self.Ho, self.Wo, self.cout, self.k, self.k,
              self.stride, self.pt, self.pl)


class ResNetTrunk(Layer):
    REPS = {"res18": [2, 2, 2, 2], "res34": [3, 4, 6, 3]}

    def __init__(self, res_type, filters, input_hw, name="resnet"):
        super().__init__(name)
        st = self.store
        if res_type not in self.REPS:
            raise NotImplementedError("%s: the reference's resnet50_/101_/152_ return a keras Model, not a tensor "
                                      "(resnet.py:217,233,249), so SAR_Net cannot use them; only res18/res34 are built" % res_type)
        if filters % 16:
            raise ValueError("res_filters must be a multiple of 16 for the gfx950 implicit-GEMM kernels (got %d)" % filters)
        stem_c = filters if res_type == "res18" else 64                      # resnet.py:173 vs 191
        if stem_c > 64 or stem_c % 4:
            raise ValueError("stem width %d unsupported (<= 64, multiple of 4)" % stem_c)
        # 2-byte trunk (csrc/conv_bf16.hip, stem_bf16.hip; compiled for bf16 and for fp16): `bf16` keeps its round-2 name and means
        # "2-byte storage"; `half` tells the two apart.  fp16 carries a static gradient scale through the trunk (include/sarnet.h)
        dt = _lib.get_trunk_dtype()
        if dt == "bf16x3" and (filters not in (32, 64) or stem_c not in (32, 64)):
            # split precision on the 2-byte kernels exists for the widths those kernels take (32: the reference's train.py:54; 64: SAR_Net's own
            # default, model.py:204); other widths run the fp32-storage kernels -- the same fp32-class arithmetic on round 1's kernels
            dt = "fp32"
            import warnings
            warnings.warn("res_filters=%d: the split-precision 2-byte conv kernels (csrc/conv_bf16.hip) take trunk widths 32 and 64 only; this model's trunk "
                          "runs the fp32-storage kernels of csrc/conv_igemm.hip instead (same fp32-class results, about twice the conv time)" % filters,
                          RuntimeWarning, stacklevel=3)
        # `x3`: split precision (csrc/trunk16.h SAR_X3): the trunk's activations / gradients are fp32 tensors in HBM (r4 v9; before: a (hi, lo)
        # pair of bf16 planes -- the same bytes), split into hi / lo bf16 operands by the conv fills; the operands only the weight gradients
        # read are single fp16 tensors (gradients times the static scale of the fp16 build)
        self.x3 = dt == "bf16x3"
        self.bf16, self.half = dt in ("bf16", "fp16", "bf16x3"), dt == "fp16"
        self.sfx, self.t16 = ("f16", torch.float16) if self.half else ("x3", torch.bfloat16) if self.x3 else ("bf16", torch.bfloat16)
        # Mixed backward of the split-precision mode (round 5; SAR_X3_BWD=x3 keeps round 4's split-precision input gradients for A/B runs): the
        # forward pass -- what north_star bounds at 1e-3 -- stays in split precision with fp32 tensors; the activation GRADIENTS are fp16 tensors
        # under the static scale and the input-gradient convs run single-pass on fp16 (sar_conv2d_f16a: the fp16 build's kernels reading the forward
        # tensors as fp32): tools/grad_recipe_emulator.py puts the flat parameter gradient at 9.6e-4 of the exact one against 6.8e-4 before (the
        # weight gradients' fp16 operands set both), at a third of the MFMAs and half the gradient bytes.
        self.bwd16 = self.x3 and os.environ.get("SAR_X3_BWD", "f16") != "x3"
        self.bsfx = "f16a" if self.bwd16 else self.sfx                       # entry points of the input-gradient launches
        self.t_grad = torch.float16 if self.bwd16 else (torch.float32 if self.x3 else self.t16)   # gradients of the residual stream in HBM
        self.wsfx = "f16" if self.x3 else self.sfx                           # build whose weight-gradient kernels run
        self.t_side = torch.float16 if self.x3 else self.t16                 # the tensors the fills write back for them
        self.t_act = torch.float32 if self.x3 else self.t16                  # activations / gradients of the residual stream in HBM
        self._gscale_cfg = _lib.get_trunk_gscale() if (self.half or self.x3) else 1.0      # 0: automatic, 64 x batch (_set_gscale)
        self.gscale = self.ginv = 1.0
        self.det = self.bf16 and _lib.get_bn_deterministic()                 # forward BatchNorm sums in 64-bit fixed point (SAR_BN_DET=1)
        if self.bf16 and not self.x3 and (filters not in (32, 64) or stem_c not in (32, 64)):
            raise ValueError("2-byte trunk: res_filters must be 32 (the reference's train.py:54 default) or 64 (SAR_Net's own default, model.py:204): "
                             "channel counts are powers of two in [32, 512]; got %d" % filters)
        self.stem = _Conv(st, st.auto_name("conv2d"), 7, 1, stem_c, 2, "same")
        self.stem_bn = _BN(st, st.auto_name("batch_normalization"), stem_c)
        self.blocks = []
        cin, f = stem_c, filters
        for i, r in enumerate(self.REPS[res_type]):
            for j in range(r):
                stride = 2 if (j == 0 and i != 0) else 1
                first = (i == 0 and j == 0)
                blk = {"cin": cin, "cout": f, "stage": i}
                blk["bn1"] = None if first else _BN(st, st.auto_name("batch_normalization"), cin)     # resnet.py:111-120
                blk["conv1"] = _Conv(st, st.auto_name("conv2d"), 3, cin, f, stride, "same")
                blk["bn2"] = _BN(st, st.auto_name("batch_normalization"), f)
                blk["conv2"] = _Conv(st, st.auto_name("conv2d"), 3, f, f, 1, "same")
                blk["sc"] = None
                if stride > 1 or cin != f:                                   # resnet.py:75-87
                    blk["sc"] = _Conv(st, st.auto_name("conv2d"), 1, cin, f, stride, "valid")
                self.blocks.append(blk)
                cin = f
            f *= 2
        self.final_bn = _BN(st, st.auto_name("batch_normalization"), cin)
        self.out_c = cin
        for cv in self._all_convs():
            cv.sfx, cv.bsfx, cv.t16, cv.epi_fwd = self.sfx, self.bsfx, self.t16, (3 if self.det else 1)
        self.plan(*input_hw)

    def _set_gscale(self, B):
        """fp16 trunk: the static gradient scale for batches of B utterances (a power of two; see _lib._TRUNK_GSCALE)."""
        s = self._gscale_cfg if self._gscale_cfg > 0 else float(2 ** int(round(np.log2(64.0 * max(B, 1)))))
        s *= 2.0 ** self.store.gscale_shift                                  # lowered by the model after a step that overflowed (engine.ParamStore)
        if s != self.gscale:
            self.gscale, self.ginv = s, 1.0 / s
            for cv in self._all_convs():
                cv.ginv = self.ginv
            self._wgplans = None                                             # the planned weight-gradient tables carry ginv

    def plan(self, T, D):
        """Spatial sizes of every layer for a (T, D) input.  Parameters do not depend on them, so an inference model
        (mode != "train": variable-length time axis, model.py:229-232) re-plans per utterance length."""
        self.in_hw = (T, D)
        H, W = self.stem.out_hw(T, D)
        self.stem_hw = (H, W)
        (H, self.pool_pt), (W, self.pool_pl) = same_pads(H, 3, 2), same_pads(W, 3, 2)
        self.pool_hw = (H, W)
        for blk in self.blocks:
            blk["in_hw"] = (H, W)
            H2, W2 = blk["conv1"].out_hw(H, W)
            blk["conv2"].out_hw(H2, W2)
            if blk["sc"] is not None:
                assert blk["sc"].out_hw(H, W) == (H2, W2)
            blk["out_hw"] = (H2, W2)
            H, W = H2, W2
        self.out_hw = (H, W)
        self.output_shape = (None, H, W, self.out_c)
        return self.out_hw

    def split_weights(self):
        """bf16 (hi, lo) planes of every eligible conv kernel, one launch (the weights change every step).  Only in the
        split-precision conv mode; the convs then stage their weight tiles with plain 16-byte copies."""
        convs = [cv for blk in self.blocks for cv in (blk["conv1"], blk["conv2"], blk["sc"]) if cv is not None and cv.wp_ok]
        on = _lib.get_conv_math() == "bf16x3" and bool(convs)
        for blk in self.blocks:
            for cv in (blk["conv1"], blk["conv2"], blk["sc"]):
                if cv is not None:
                    cv.use_planes = on and cv.wp_ok
        if not on:
            return
        if getattr(self, "_wtable", None) is None:
            rows, blk0 = [], 0
            for cv in convs:
                r = cv.planes()
                rows.append(r + [blk0]); blk0 += (r[5] // 32) * (r[6] // 64)
            self._wtable = torch.tensor(rows, dtype=torch.int64, device="cuda")
            self._wblocks = blk0
        _call("sar_conv_weight_split_bf16", self._wtable, self._wtable.shape[0], self._wblocks)

    # ------------------------------------------------------------------ bf16 trunk (csrc/conv_bf16.hip)
    def _all_convs(self):
        return [cv for blk in self.blocks for cv in (blk["conv1"], blk["conv2"], blk["sc"]) if cv is not None]

    def _all_bns(self):
        return [self.stem_bn] + [bn for blk in self.blocks for bn in (blk["bn1"], blk["bn2"]) if bn is not None] + [self.final_bn]

    def _setup16(self):
        if getattr(self, "_pack_table", None) is not None:
            return
        rows, blk0 = [], 0
        for cv in self._all_convs():
            r, nb = cv.streams16()
            rows.append(r + [blk0, 1 if self.bwd16 else 0]); blk0 += nb     # (bit 0: the input-gradient stream as fp16 fragments, csrc/conv_bf16.hip: PackRow)
        self._pack_table = torch.tensor(rows, dtype=torch.int64, device="cuda")
        self._pack_blocks = blk0
        bns = self._all_bns()
        n = sum((6 if self.det else 4) * _lib.bn_bins() * bn.C for bn in bns)
        self._arena = self.store.bufs.get(self.name + "/bn_arena", (n,))
        off = 0
        for bn in bns:
            off = bn.prepare16(self._arena, off, self.det)

    def _buf16(self, key, shape):
        """An activation / gradient tensor of the trunk: the build's 2-byte type; split precision: fp32."""
        return self.buf(key, shape, self.t_act)

    def _gbuf16(self, key, shape):
        """A gradient tensor of the residual stream (mixed backward: fp16 under the static scale; else the trunk's activation type)."""
        return self.buf(key, shape, self.t_grad)

    def _side16(self, key, shape):
        """An operand only the weight gradients read (the fills write it back): the trunk's 2-byte type; split precision: ONE fp16 tensor."""
        return self.buf(key, shape, self.t_side)

    def _stem_fn(self, name):
        return name + ("_f16" if self.half else "_x3" if self.x3 else "")

    def _forward16(self, x, B):
        """Same graph as forward() on 2-byte activations: one launch per conv (BN+ReLU of the producer evaluated in the conv's
        patch fill from the statistic bins; statistics of the consumer BN out of the epilogue), the stem in recompute form."""
        self._setup16()
        if self.half or self.x3:
            self._set_gscale(B)
        training = self.store.training
        ztab = self.store.take_deferred_zero()                              # the step-start zeroing rides in the packing launch (model._device_step)
        if ztab is not None:
            _call("sar_conv_pack_weights_zero_" + self.sfx, self._pack_table, self._pack_table.shape[0], self._pack_blocks, ztab, ztab.shape[0])
        else:
            _call("sar_conv_pack_weights_" + self.sfx, self._pack_table, self._pack_table.shape[0], self._pack_blocks)
        if training:
            if not self.store.zero_at_step_start(self._arena):              # the step-start launch covers it from the second step on
                _call("sar_fill_f32", self._arena, 0.0, self._arena.numel())
        else:
            for bn in self._all_bns():
                bn.fold16()
        st = "fwd" if training else "saved"
        cur = self._stem_forward16(x, B, st)
        nblk = len(self.blocks)
        for n, blk in enumerate(self.blocks):
            H, W = blk["in_hw"]; H2, W2 = blk["out_hw"]
            cin, cout = blk["cin"], blk["cout"]
            blk["x"] = cur
            bn1, bn2 = blk["bn1"], blk["bn2"]
            c1 = self._buf16("b%d_c1" % n, (B, H2, W2, cout))
            # training: the activated operands relu(bn(.)) the fills compute are written back (a, a1) for the weight gradients
            blk["a"] = self._side16("b%d_a" % n, (B, H, W, cin)) if (training and bn1 is not None) else None
            blk["a1"] = self._side16("b%d_a1" % n, (B, H2, W2, cout)) if training else None
            # split precision: the RAW block input as the weight gradients read it (the shortcut's, the first conv's): one fp16 tensor, written
            # by the fill of the shortcut conv (fill 0 + side = a copy); the first block's comes out of the stem's pooling pass
            blk["x16"] = cur
            if self.x3 and training and (bn1 is None or blk["sc"] is not None):
                blk["x16"] = self.pooled_h if n == 0 else self._side16("b%d_x16" % n, (B, H, W, cin))
            a1 = blk["conv1"].fwd16_args(cur, c1, B, fill=1 if bn1 is not None else 0, fbn=bn1.ref(B * H * W, st, training) if bn1 is not None else None,
                                         e_bins=bn2.bins if training else None, side=blk["a"])
            if blk["sc"] is not None:
                # the block's first conv and its 1x1 shortcut read the same tensor and write different ones: ONE launch where their
                # instantiations pair up (sar_conv2d_pair_*), else one after the other
                sc = self._buf16("b%d_sc" % n, (B, H2, W2, cout))
                a2 = blk["sc"].fwd16_args(cur, sc, B, side=blk["x16"] if (self.x3 and training and n > 0) else None)
                if not self._pair(a1, a2):
                    _call("sar_conv2d_" + self.sfx, *a1); _call("sar_conv2d_" + self.sfx, *a2)
            else:
                _call("sar_conv2d_" + self.sfx, *a1)
                sc = cur
            y = self._buf16("b%d_y" % n, (B, H2, W2, cout))
            nxt = self.blocks[n + 1]["bn1"] if n + 1 < nblk else self.final_bn
            blk["conv2"].fwd16(c1, y, B, fill=1, fbn=bn2.ref(B * H2 * W2, st, training), res=sc, e_bins=nxt.bins if training else None,
                               side=blk["a1"])
            blk["c1"] = c1
            cur = y
        H, W = self.out_hw
        self.pre_final = cur
        out = self.buf("out", (B, H, W, self.out_c))
        _call("sar_bn_apply_%s_f32" % self.sfx, cur, self.final_bn.ref(B * H * W, st, training), out, B * H * W, self.out_c, 1)
        return out

    def _backward16(self, dy, part):
        B = self.B
        x3 = self.x3
        # split precision: the gradient stream (g, da, dsc: fp32 tensors) is in TRUE units; only the fp16 copies the weight gradients read
        # (g16, dc1) carry the static scale, which their descriptors' ginv undoes.  fp16 build: the whole stream is scaled.
        hyb = self.bwd16                                          # mixed backward: the fp16 build's gradient stream (scaled) beside the fp32 forward tensors
        gi = 1.0 if (x3 and not hyb) else self.ginv
        tg = self.t_grad
        if part in (None, 0, "01"):
            H, W = self.out_hw
            fb = self.final_bn
            R = B * H * W
            g16 = self.scratch("res16_g16_%dx%dx%d" % (H, W, self.out_c), (B, H, W, self.out_c), self.t_side) if x3 else None
            g = g16 if hyb else self.scratch("res16_g_%dx%dx%d" % (H, W, self.out_c), (B, H, W, self.out_c), self.t_act)
            g16 = g if g16 is None else g16
            # the gradient enters the trunk here: fp32 sums and constants, the result times the fp16 build's gradient scale (1 for bf16)
            _call("sar_bn_bwd_reduce_" + self.sfx, dy, 1, self.pre_final, fb.saved, fb.bbins, R, self.out_c, 1)
            if x3:
                _call("sar_bn_bwd_apply_x3", dy, 1, self.pre_final, fb.ref(R, "bwd", True), None, None if hyb else g, g16, R, self.out_c, 1, self.gscale)
            else:
                _call("sar_bn_bwd_apply_" + self.sfx, dy, 1, self.pre_final, fb.ref(R, "bwd", True), None, g, R, self.out_c, 1, self.gscale)
        else:
            g, g16 = self._g_part
        stages = {None: (3, 2, 1, 0), 0: (3,), 1: (2,), 2: (1, 0), "01": (3, 2)}[part]
        bf = self.t16
        self._wg16 = []
        overlap = part is None and self.WGRAD_OVERLAP
        if overlap and getattr(self, "_wg_stream", None) is None:
            self._wg_stream = torch.cuda.Stream()
        # `pend`: the gradient the block behind handed over, not materialised yet -- (da1, x, shortcut gradient, bn1 of that block):
        # g = bn1_bwd(da1, x) + dsc is formed in the fill of THIS block's conv2 input-gradient launch (fill 3) and written back to
        # `g` for the gradient's other consumers (weight gradients, the shortcut path), instead of by a BatchNorm-apply launch
        pend = self._pend16 if part not in (None, 0, "01") else None
        tap = getattr(self, "debug_tap", None)
        for n in range(len(self.blocks) - 1, -1, -1):
            blk = self.blocks[n]
            if blk["stage"] not in stages:
                continue
            H, W = blk["in_hw"]; H2, W2 = blk["out_hw"]
            cin, cout = blk["cin"], blk["cout"]
            R1, R2 = B * H * W, B * H2 * W2
            bn1, bn2, c1, x = blk["bn1"], blk["bn2"], blk["c1"], blk["x"]
            # conv2: operand a1 = relu(bn2(c1)) (written back by the forward fill); its input gradient da2 leaves the two sums of
            # bn2's backward in bn2.bbins
            red = []
            da2 = self.scratch("res16_da_%dx%dx%d" % (H2, W2, cout), (B, H2, W2, cout), tg)
            if pend is None:
                blk["conv2"].dgrad16(g, None, da2, B, e_bins=bn2.bbins, e_y=c1, e_saved=bn2.saved)
            else:
                p_da1, p_x, p_dsc, p_bn, p_R, p_tap = pend
                if x3 and not hyb:                                    # g as fp32 (side2) for the convs, as scaled fp16 (side) for the weight gradients
                    blk["conv2"].dgrad16(p_da1, p_x, da2, B, fill=3, fbn=p_bn.ref(p_R, "bwd", True, gi), dy3=p_dsc, side=g16, side2=g,
                                         side_scale=self.gscale, e_bins=bn2.bbins, e_y=c1, e_saved=bn2.saved)
                else:
                    blk["conv2"].dgrad16(p_da1, p_x, da2, B, fill=3, fbn=p_bn.ref(p_R, "bwd", True, gi), dy3=p_dsc, side=g,
                                         e_bins=bn2.bbins, e_y=c1, e_saved=bn2.saved)
                if p_tap is not None:                                 # tests: (block, its g, its da2 copy) + the now materialised gin
                    tap(p_tap[0], p_tap[1], p_tap[2], g)
                pend = None
            red.append(blk["conv2"].wgrad16(blk["a1"], g16, B, bias_grad=True))
            # the weight gradients run at the end of the part (_wgrad16), so their gradient operands -- the block's incoming
            # gradient and dc1 -- live in per-block buffers, not in recycled scratch
            gin = self._gbuf16("b%d_gin" % n, (B, H, W, cin))
            gin16 = self._side16("b%d_gin16" % n, (B, H, W, cin)) if (x3 and not hyb and bn1 is not None) else gin
            sc_args = None
            if blk["sc"] is not None:
                dsc = gin if bn1 is None else self._gbuf16("b%d_dsc" % n, (B, H, W, cin))
                sc_args = blk["sc"].dgrad16_args(g, None, dsc, B)
                if bn1 is None:                                       # (conv1's input gradient below adds dsc: one after the other)
                    _call("sar_conv2d_" + self.bsfx, *sc_args); sc_args = None
                red.append(blk["sc"].wgrad16(blk["x16"], g16, B, bias_grad=True))
            else:
                dsc = g
            # conv1's output gradient dc1 = BN+ReLU backward of (da2, c1): evaluated in the fill of conv1's input-gradient launch
            # (which owns bn2's parameter gradients) and written back for the weight gradient
            dc1 = self._side16("b%d_dc1" % n, (B, H2, W2, cout))
            if bn1 is not None:
                da1 = self._gbuf16("b%d_da1" % n, (B, H, W, cin))
                c1_args = blk["conv1"].dgrad16_args(da2, c1, da1, B, fill=2, fbn=bn2.ref(R2, "bwd", True, gi), e_bins=bn1.bbins, e_y=x, e_saved=bn1.saved,
                                                    side=dc1, side_scale=self.gscale)
                # the input gradients of the first conv and of the shortcut are independent (their sum is formed in the fill of the block in
                # front, fill 3): one launch where they pair up
                if sc_args is None or not self._pair(c1_args, sc_args, self.bsfx):
                    if sc_args is not None:
                        _call("sar_conv2d_" + self.bsfx, *sc_args)
                    _call("sar_conv2d_" + self.bsfx, *c1_args)
                pend = (da1, x, dsc, bn1, R1, (n, g, da2.clone()) if tap is not None else None)
            else:
                # first block of the first stage: conv1 consumes the pooled tensor directly (resnet.py:111-117)
                blk["conv1"].dgrad16(da2, c1, gin, B, fill=2, fbn=bn2.ref(R2, "bwd", True, gi), res=dsc, side=dc1, side_scale=self.gscale)
                if tap is not None:
                    tap(n, g, da2, gin)
            red.append(blk["conv1"].wgrad16(blk["a"] if bn1 is not None else blk["x16"], dc1, B, bias_grad=False))
            self._wg16.extend(red)
            g, g16 = gin, gin16
            # whole backward in one call: a finished stage's weight gradients start on the side stream while the next stage's
            # input-gradient chain (short, latency-bound launches that leave most of every CU idle) runs on the main stream
            if overlap and (n == 0 or self.blocks[n - 1]["stage"] != blk["stage"]):
                self._wgrad16(("stage", blk["stage"]), side=True)
                self._wg16 = []
        self._pend16 = pend
        assert pend is None or part in (0, 1, "01"), "the first block's conv1 takes the pooled tensor directly (resnet.py:111-117)"
        self._g_part = (g, g16)
        if not overlap:
            self._wgrad16(part)
        if part in (0, 1, "01"):
            return None
        self._stem_backward16(g, B)
        if overlap:
            torch.cuda.current_stream().wait_stream(self._wg_stream)       # the branch joins in front of the optimiser
        return None

    # launch classes of the batched weight gradient: (kernel size, stride or None, 64-wide channel tiles) -> blocks the planner
    # aims for.  3x3 stride 1 with Cin, Cout >= 64 (stages 1-3): 64 x 64 tiles, one 9-wave block per CU; the 32-channel layers:
    # 32 x 32 tiles, three blocks per CU; stride-2 convs need a 4x larger patch ring, so they get their own launch.
    # (targets re-measured at v36 with the flush known to be 40 % of the 64-wide launch -- every item ends in 147 KB of fp32 atomics,
    # tools/conv_variants.py phase removal: 156 us with, 94 without --: 256 / 320 / 384 / 512 / 640 / 896 items = 163 / 135 / 139 / 166 /
    # 156 / 178 us; 32-wide class 256 / 384 / 512 / 768 / 1024 = 133 / 132 / 116 / 123 / 129 us)
    WGRAD_CLASSES = (((3, 1, True), int(os.environ.get("SAR_WG_T64", 448))), ((3, 1, False), int(os.environ.get("SAR_WG_T32", 512))),
                     ((3, 2, False), 512), ((1, None, False), 256))

    @staticmethod
    def _wgrad_class(d):
        if d["K"] == 1:
            return (1, None, False)
        return (3, d["s"], d["s"] == 1 and d["Cin"] >= 64 and d["Cout"] >= 64)

    # SAR_WGRAD_OVERLAP=1: per-stage weight-gradient launches on a second stream underneath the next stage's input-gradient
    # chain.  Measured slower (cfg2: 3.14 vs 2.90 ms per step): the batched launch is bound by the memory system and takes it
    # away from the chain's fills; default = one batch at the end of the backward call, on the main stream.
    WGRAD_OVERLAP = os.environ.get("SAR_WGRAD_OVERLAP", "0") == "1"
    STEM_X3 = os.environ.get("SAR_STEM_X3", "1") != "0"           # split precision: the recompute-form stem (0: round 1's fp32 stem + casts)
    PAIR = os.environ.get("SAR_CONV_PAIR", "1") != "0"

    def _pair(self, a3x3, b1x1, sfx=None):
        """Both convolutions (argument tuples of sar_conv2d_*) in one launch; False: nothing was launched."""
        if not self.PAIR:
            return False
        return _lib.conv_pair(_lib.ConvArgs.of(*a3x3), _lib.ConvArgs.of(*b1x1), sfx or self.sfx)

    def _wgrad16(self, part, side=False):
        """The weight gradients of every conv collected in self._wg16 (a backward part, or one stage): one launch per class,
        tables planned once per (B, T).  side: on the trunk's second stream, forked from the main stream here (inside a HIP-graph
        capture this becomes a parallel branch); _backward16 joins it."""
        plans = getattr(self, "_wgplans", None)
        if plans is None:
            plans = self._wgplans = {}
        key = (part, self.B, self.in_hw, tuple((d["x"].data_ptr(), d["dy"].data_ptr()) for d in self._wg16))
        if key not in plans:
            ps = []
            for cls, target in self.WGRAD_CLASSES:
                ds = [d for d in self._wg16 if self._wgrad_class(d) == cls]
                if ds:
                    ps.append(_lib.WgradPlan(ds, target, (2, 2) if cls[2] else (1, 1), self.wsfx))
            plans[key] = ps
        if side:
            self._wg_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._wg_stream):
                for p in plans[key]:
                    p.launch()
        else:
            # the partial tiles of all classes are summed by ONE launch behind the last batch (a launch costs ~4.5 us of the step before it
            # does anything: DESIGN 5.0); SAR_WG_REDUCE_MULTI=0: one reduction behind every batch
            multi = os.environ.get("SAR_WG_REDUCE_MULTI", "1") != "0" and len(plans[key]) > 1
            if multi and os.environ.get("SAR_WG_FORK", "0") == "1":
                # (EXPERIMENT: the classes are independent of each other -- the first on this stream, the rest on a second one, joined in
                # front of the reduction; inside a HIP-graph capture a parallel branch)
                main = torch.cuda.current_stream()
                if getattr(self, "_wg_stream", None) is None:
                    self._wg_stream = torch.cuda.Stream()
                self._wg_stream.wait_stream(main)
                plans[key][0].launch(defer_reduce=True)
                with torch.cuda.stream(self._wg_stream):
                    for p in plans[key][1:]:
                        p.launch(defer_reduce=True)
                main.wait_stream(self._wg_stream)
                _lib.WgradPlan.reduce_multi(plans[key], self.wsfx)
                return
            for p in plans[key]:
                p.launch(defer_reduce=multi)
            if multi:
                # in a train step the kernel gradients are zero here (zeroed at the step's start, this reduction their only writer): written, not added
                _lib.WgradPlan.reduce_multi(plans[key], self.wsfx, overwrite=self.store.__dict__.get("_zeroed_this_step", False)
                                            and os.environ.get("SAR_WG_REDUCE_WRITE", "1") == "1")

    # stem of the bf16 trunk (csrc/stem_bf16.hip): the 7x7 conv is recomputed wherever its output is needed, so only the pooled map
    # (bf16) and its argmax slots exist in HBM -- no (B, T/2, D/2, C0) tensor, forward or backward
    def _stem_geo(self, B):
        T, D = self.in_hw
        (H, W), (Hp, Wp) = self.stem_hw, self.pool_hw
        return (B, T, D, H, W, self.stem.cout, self.stem.pt, self.stem.pl), (Hp, Wp, self.pool_pt, self.pool_pl)

    def _stem_forward16(self, x, B, st):
        if self.x3 and not self.STEM_X3:
            # (SAR_STEM_X3=0, for A/B runs) round 1's fp32 stem (full-resolution map), its pooled map copied + the fp16 copy the first
            # block's weight gradients read
            p32 = self._stem_forward(x, B)
            Hp, Wp = self.pool_hw; c0 = self.stem.cout
            self.pooled16 = self._buf16("pooled16", (B, Hp, Wp, c0)); self.pooled_h = self._side16("pooled_h", (B, Hp, Wp, c0))
            _call("sar_cast_f32_x3", p32, self.pooled16, self.pooled_h, 1.0, p32.numel())
            return self.pooled16
        conv, pool = self._stem_geo(B)
        H, W = self.stem_hw; Hp, Wp = self.pool_hw; c0 = self.stem.cout
        sb, cv = self.stem_bn, self.stem
        training = self.store.training
        if training:
            _call(self._stem_fn("sar_stem16_stats"), x, cv.kernel.w, cv.bias.w, sb.bins, *conv, int(self.det))
        self.pooled16 = self._buf16("pooled16", (B, Hp, Wp, c0)); self.pool_idx = self.buf("pool_idx", (B, Hp, Wp, c0), torch.uint8)
        if self.x3:                                                          # + the pooled map as one fp16 tensor: what the first block's weight gradients read
            self.pooled_h = self._side16("pooled_h", (B, Hp, Wp, c0))
            _call("sar_stem16_pool_fwd_x3", x, cv.kernel.w, cv.bias.w, sb.ref(B * H * W, st, training), self.pooled16, self.pooled_h if training else None,
                  self.pool_idx, *conv, *pool)
            return self.pooled16
        _call(self._stem_fn("sar_stem16_pool_fwd"), x, cv.kernel.w, cv.bias.w, sb.ref(B * H * W, st, training), self.pooled16, self.pool_idx, *conv, *pool)
        return self.pooled16

    def _stem_backward16(self, g, B):
        """g: gradient of the pooled map (bf16).  BatchNorm sums from (g, pooled, slots), then the 7x7 kernel gradient."""
        if self.x3 and not self.STEM_X3:
            if self.bwd16:
                # the mixed backward hands this function ONE fp16 tensor under the gradient scale; round 1's fp32 stem takes a true-unit fp32
                # gradient and sar_cast_x3_f32 would read the 2-byte tensor as 4-byte elements
                raise RuntimeError("SAR_STEM_X3=0 (round 1's fp32 stem, an A/B switch) needs SAR_X3_BWD=x3: the default mixed backward "
                                   "carries the stem's gradient as a scaled fp16 tensor")
            g32 = self.scratch("stem_g32", tuple(g.shape[1:]))
            _call("sar_cast_x3_f32", g, g32, g32.numel())
            return self._stem_backward(g32, B)
        conv, pool = self._stem_geo(B)
        H, W = self.stem_hw
        sb, cv = self.stem_bn, self.stem
        if self.x3 and not self.bwd16:
            # (SAR_X3_BWD=x3, A/B runs: the fp32 true-unit gradient of round 4's backward as the scaled fp16 tensor the stem kernels take since round 5)
            g16 = self.scratch("stem_g16", tuple(g.shape), torch.float16); gcp = self.scratch("stem_gcp", tuple(g.shape), torch.float32)
            _call("sar_cast_f32_x3", g, gcp, g16, self.gscale, g.numel())
            g = g16
        ref = sb.ref(B * H * W, "bwd", True, self.ginv)
        work = self.scratch("stem16_work", (_lib.stem16_wgrad_workspace_bytes(cv.cout) // 4,))
        if self.x3 and self.bwd16 and getattr(self, "pooled_h", None) is not None and os.environ.get("SAR_STEM_RED16", "1") == "1":
            # the BatchNorm-backward sums from the fp16 copy of the pooled map (half the bytes of this HBM-bound pass)
            _call("sar_stem16_bwd_reduce_x3h", g, self.pooled_h, self.pool_idx, self.x, cv.kernel.w, cv.bias.w, ref, work, *conv, *pool)
        else:
            _call(self._stem_fn("sar_stem16_bwd_reduce"), g, self.pooled16, self.pool_idx, self.x, cv.kernel.w, cv.bias.w, ref, work, *conv, *pool)
        _call(self._stem_fn("sar_stem16_wgrad"), self.x, cv.kernel.w, cv.bias.w, g, self.pooled16, self.pool_idx, ref, cv.kernel.g, work, *conv, *pool)

    def _stem_forward(self, x, B):
        T, D = self.in_hw
        H, W = self.stem_hw
        c0 = self.stem.cout
        self.stem_out = self.buf("stem_out", (B, H, W, c0))
        sb = self.stem_bn
        if self.store.training:                                              # statistics out of the conv epilogue
            sb.prepare()
            _call("sar_stem_conv7x7_fwd_bnstats_f32", x, self.stem.kernel.w, self.stem.bias.w, self.stem_out, B, T, D, H, W, c0,
                  self.stem.pt, self.stem.pl, sb.gamma.w, sb.beta.w, sb.sums, sb.mean, sb.rstd, sb.scale, sb.shift,
                  sb.moving_mean.w, sb.moving_var.w, BN_EPS, BN_MOMENTUM, 1)
        else:
            _call("sar_stem_conv7x7_fwd_f32", x, self.stem.kernel.w, self.stem.bias.w, self.stem_out, B, T, D, H, W, c0, self.stem.pt, self.stem.pl)
            sb.stats(self.stem_out, B * H * W)
        Hp, Wp = self.pool_hw
        self.pooled = self.buf("pooled", (B, Hp, Wp, c0)); self.pool_idx = self.buf("pool_idx", (B, Hp, Wp, c0), torch.uint8)
        _call("sar_bn_relu_maxpool_fwd_f32", self.stem_out, self.stem_bn.scale, self.stem_bn.shift, self.pooled, self.pool_idx,
              B, H, W, c0, Hp, Wp, self.pool_pt, self.pool_pl)
        return self.pooled

    def _stem_backward(self, g, B):
        """maxpool + BN + ReLU backward, then the 7x7 weight gradient (no input gradient needed); g fp32 (B,Hp,Wp,c0)."""
        Hs, Ws = self.stem_hw; Hp, Wp = self.pool_hw; c0 = self.stem.cout
        sb = self.stem_bn
        # the full-resolution gradient (131 MB at B=64) is never written: the two BatchNorm sums first (dx = None), then the
        # weight gradient evaluates the pooling/BatchNorm backward tile by tile
        _call("sar_maxpool_bn_relu_bwd_f32", g, self.pooled, self.pool_idx, self.stem_out, sb.mean, sb.rstd, sb.scale, sb.shift, sb.sums,
              None, B, Hs, Ws, c0, Hp, Wp, self.pool_pt, self.pool_pl)
        _call("sar_bn_param_grads_f32", sb.sums, sb.gamma.g, sb.beta.g, c0)
        T, D = self.in_hw
        # the stem bias feeds a BatchNorm: its gradient is exactly zero, so db is not computed (None)
        _call("sar_stem_conv7x7_wgrad_poolbwd_f32", self.x, self.stem_out, g, self.pooled, self.pool_idx, sb.mean, sb.rstd, sb.scale,
              sb.sums, self.stem.kernel.g, None, B, T, D, Hs, Ws, c0, self.stem.pt, self.stem.pl, Hp, Wp, self.pool_pt, self.pool_pl)

    # ------------------------------------------------------------------ forward
    def forward(self, x, B):
        self.B, self.x = B, x
        if self.bf16:
            return self._forward16(x, B)
        self.split_weights()
        self._stem_forward(x, B)
        c0 = self.stem.cout
        cur = self.pooled
        # training: every conv also produces the batch statistics of the BatchNorm that consumes its output (no separate
        # pass over the activation); inference: the moving averages are folded instead (stats() does that)
        fused = self.store.training
        nblk = len(self.blocks)
        for n, blk in enumerate(self.blocks):
            H, W = blk["in_hw"]; H2, W2 = blk["out_hw"]
            blk["x"] = cur
            if blk["bn1"] is not None and not (fused and n > 0):
                blk["bn1"].stats(cur, B * H * W)
            c1 = self.buf("b%d_c1" % n, (B, H2, W2, blk["cout"]))
            blk["conv1"].fwd(cur, c1, B, bn=blk["bn1"], bn_out=blk["bn2"] if fused else None)
            if not fused:
                blk["bn2"].stats(c1, B * H2 * W2)
            if blk["sc"] is not None:
                sc = self.buf("b%d_sc" % n, (B, H2, W2, blk["cout"]))
                blk["sc"].fwd(cur, sc, B)
            else:
                sc = cur
            y = self.buf("b%d_y" % n, (B, H2, W2, blk["cout"]))
            nxt = self.blocks[n + 1]["bn1"] if n + 1 < nblk else self.final_bn
            blk["conv2"].fwd(c1, y, B, bn=blk["bn2"], residual=sc, bn_out=nxt if fused else None)
            blk["c1"] = c1
            cur = y
        H, W = self.out_hw
        self.pre_final = cur
        if not fused:
            self.final_bn.stats(cur, B * H * W)
        out = self.buf("out", (B, H, W, self.out_c))
        _call("sar_bn_apply_f32", cur, self.final_bn.scale, self.final_bn.shift, out, B * H * W, self.out_c, 1)
        return out

    # ------------------------------------------------------------------ backward
    # Backward runs in parts so that a data-parallel step can all-reduce the gradients of the late stages (most of the
    # parameters: stage 3 = 14 MB, stage 2 = 7 MB of the 21 MB conv bucket at F=32) underneath the backward of the early ones:
    # part 0 = final BN + stage 3, part 1 = stage 2, part 2 = stages 1, 0 and the stem.  The flat parameter buffer is in
    # forward order, so each part completes one contiguous range (bucket_split_names()).
    N_PARTS = 3

    def bucket_split_names(self):
        """Where the gradient buckets of a data-parallel step are cut (creation order = flat-buffer order): at conv1's KERNEL of the
        first block of stage 3 and of stage 2 -- not at that block's bn1.  In the bf16 trunk the BatchNorm backward of a stage's first
        block is evaluated in the fill of the PREVIOUS block's conv2 input-gradient launch (`pend` in _backward16), i.e. its gamma / beta
        gradients are written by the NEXT backward part; with the cut at conv1.kernel they belong to that part's bucket, so no bucket is
        written after its all-reduce has been launched (the cut at bn1.gamma raced with the in-flight collective and dropped
        the other ranks' contribution)."""
        return [next(b for b in self.blocks if b["stage"] == stage)["conv1"].kernel.name for stage in (3, 2)]

    def late_bn_params(self):
        """Names of the parameters whose gradients the bf16 trunk writes one backward part later than their block's (see above)."""
        out = []
        for stage in (3, 2):
            bn1 = next(b for b in self.blocks if b["stage"] == stage)["bn1"]
            if bn1 is not None:
                out += [bn1.gamma.name, bn1.beta.name]
        return out

    def backward(self, dy, part=None):
        """part None: the whole backward; 0/1/2: one part (see above), "01": parts 0 and 1 in one call; dy only read by part 0 / "01"."""
        if self.bf16:
            return self._backward16(dy, part)
        B = self.B
        if part in (None, 0, "01"):
            H, W = self.out_hw
            g = self.scratch("res_g_%dx%dx%d" % (H, W, self.out_c), (B, H, W, self.out_c))
            self.final_bn.bwd(dy, self.pre_final, None, g, B * H * W, relu=1)
        else:
            g = self._g_part
        stages = {None: (3, 2, 1, 0), 0: (3,), 1: (2,), 2: (1, 0), "01": (3, 2)}[part]
        for n in range(len(self.blocks) - 1, -1, -1):
            blk = self.blocks[n]
            if blk["stage"] not in stages:
                continue
            H, W = blk["in_hw"]; H2, W2 = blk["out_hw"]
            cin, cout = blk["cin"], blk["cout"]
            R2 = B * H2 * W2
            # conv2: operand = relu(bn2(c1))
            da2 = self.scratch("res_da_%dx%dx%d" % (H2, W2, cout), (B, H2, W2, cout))
            blk["conv2"].dgrad(g, da2, B)
            blk["conv2"].wgrad(blk["c1"], g, B, bn=blk["bn2"])
            dc1 = da2                                                      # in place: BN+ReLU backward
            blk["bn2"].bwd(da2, blk["c1"], None, dc1, R2, relu=1)
            blk["conv1"].wgrad(blk["x"], dc1, B, bn=blk["bn1"], bias_grad=False)   # feeds bn2
            gin = self.scratch("res_gin%d_%dx%dx%d" % (n & 1, H, W, cin), (B, H, W, cin))   # ping-pong: never aliases g
            if blk["bn1"] is not None:
                # shortcut gradient first (raw block input), then the main path folds it in via `add`
                if blk["sc"] is not None:
                    dsc = self.scratch("res_dsc_%dx%dx%d" % (H, W, cin), (B, H, W, cin))
                    blk["sc"].dgrad(g, dsc, B); blk["sc"].wgrad(blk["x"], g, B)
                else:
                    dsc = g
                da1 = self.scratch("res_da1_%dx%dx%d" % (H, W, cin), (B, H, W, cin))
                blk["conv1"].dgrad(dc1, da1, B)
                blk["bn1"].bwd(da1, blk["x"], dsc, gin, B * H * W, relu=1)
            else:
                # first block of the first stage: conv1 consumes the pooled tensor directly (resnet.py:111-117)
                if blk["sc"] is not None:
                    blk["sc"].dgrad(g, gin, B); blk["sc"].wgrad(blk["x"], g, B)
                    blk["conv1"].dgrad(dc1, gin, B, accumulate=1)
                else:
                    blk["conv1"].dgrad(dc1, gin, B)
                    _call("sar_axpy_f32", g, gin, 1.0, gin.numel())
            g = gin
        self._g_part = g
        if part in (0, 1, "01"):
            return None
        self._stem_backward(g, B)
        return None


def _make(res_type, input, filters):
    shape = getattr(input, "shape", input)
    T, D = int(shape[-3]), int(shape[-2])
    return ResNetTrunk(res_type, filters, (T, D))


def resnet18_(input, filters=64):
    """resnet.py:170-182.  `input`: an InputSpec / shape (T, D, 1).  Returns the trunk layer (call .forward)."""
    return _make("res18", input, filters)


def resnet34_(input, filters=64):
    """resnet.py:188-201 (64-channel stem regardless of `filters`)."""
    return _make("res34", input, filters)


def resnet50_(input, filters=64):
    return _make("res50", input, filters)


def resnet101_(input, filters=64):
    return _make("res101", input, filters)


def resnet152_(input, filters=64):
    return _make("res152", input, filters)
