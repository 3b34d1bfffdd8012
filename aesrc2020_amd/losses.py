"""Margin-softmax heads and circle loss, MI355X-native (mirrors /root/reference/losses.py).

SphereFace / CosFace / ArcFace keep the reference constructor (n_classes, s, m, regularizer) and call convention
`layer([x, y_onehot]) -> softmax probabilities` (losses.py:28-47, 74-91, 121-144); weight `W (D, n_classes)`,
glorot_uniform (losses.py:22-26).  circle_loss(y_true, y_pred, gamma, margin) -> per-sample loss (losses.py:157-172).
"""
import numpy as np

from . import _lib
from .engine import glorot_uniform
from .layers import Layer

_call = _lib.call


class _MarginHead(Layer):
    MODE = None

    def __init__(self, n_classes=10, s=30.0, m=None, regularizer=None, name=None, in_dim=None, **kwargs):
        super().__init__(name or self.__class__.__name__.lower())
        self.n_classes, self.s, self.m = int(n_classes), float(s), float(self.DEFAULT_M if m is None else m)
        self.l2 = float(regularizer) if regularizer else 0.0      # reference passes regularizer=None
        self.W = None
        if in_dim is not None:
            self.build(in_dim)

    def build(self, in_dim):
        A = self.n_classes
        self.W = self.store.add(self.name + "/W", (in_dim, A), glorot_uniform(self.store.rng, (in_dim, A), in_dim, A), self.l2)
        self.D = in_dim
        self.output_shape = (None, A)

    def compute_output_shape(self, input_shape):
        return (None, self.n_classes)

    def __call__(self, inputs):
        return self.forward(*inputs)

    def forward(self, x, y):
        B, D, A = x.shape[0], self.D, self.n_classes
        self.x, self.y, self.B = x, y, B
        self.probs = self.buf("probs", (B, A)); self.cosv = self.buf("cos", (B, A))
        self.inv_x = self.buf("inv_x", (B,)); self.inv_w = self.buf("inv_w", (A,))
        _call("sar_margin_head_fwd_f32", x, y, self.W.w, self.probs, self.cosv, self.inv_x, self.inv_w, B, D, A, self.MODE, self.s, self.m)
        return self.probs

    def train(self, x, y, y_true, dloss, dx_out, accumulate_dx=False):
        """forward + categorical_crossentropy(y_true, probs) + backward of both in two launches (csrc/heads.hip: margin_train_kernel):
        -> (probs, per-sample loss); dx_out written / accumulated, W.g accumulated."""
        B, D, A = x.shape[0], self.D, self.n_classes
        self.x, self.y, self.B = x, y, B
        self.probs = self.buf("probs", (B, A)); self.inv_w = self.buf("inv_w", (A,)); loss = self.buf("loss", (B,))
        key = self.name + "/dwn_ws"
        fresh = key not in self.store.bufs.d
        ws = self.buf("dwn_ws", (D, A))
        if fresh:
            ws.zero_()                                             # the kernels keep it zero from here on (first step is eager)
        _call("sar_margin_head_train_f32", x, y, y_true, dloss, self.W.w, self.probs, loss, self.inv_w, dx_out, self.W.g, ws, B, D, A,
              self.MODE, self.s, self.m, 1 if accumulate_dx else 0)
        return self.probs, loss

    def train_buffers(self, x, y, B):
        """Buffers of the training step when another launch runs it (model.py: sar_accent_margin_train_f32): -> (workspace of D*A + 1
        words the kernel keeps zero, probs, per-sample loss, inv_w)."""
        D, A = self.D, self.n_classes
        self.x, self.y, self.B = x, y, B
        self.probs = self.buf("probs", (B, A)); self.inv_w = self.buf("inv_w", (A,))
        key = self.name + "/dwn_ws1"
        fresh = key not in self.store.bufs.d
        ws = self.buf("dwn_ws1", (D * A + 1,))
        if fresh:
            ws.zero_()
        return ws, self.probs, self.buf("loss", (B,)), self.inv_w

    def backward(self, dprobs, dx_out=None, accumulate_dx=False):
        B, D, A = self.B, self.D, self.n_classes
        dx = dx_out if dx_out is not None else self.buf("dx", (B, D))
        ws = self.buf("dwn_ws_bwd", (D, A))
        _call("sar_margin_head_bwd_f32", dprobs, self.probs, self.cosv, self.x, self.y, self.W.w, self.inv_x, self.inv_w, dx,
              self.W.g, ws, B, D, A, self.MODE, self.s, self.m, 1 if accumulate_dx else 0)
        return dx


class SphereFace(_MarginHead):
    MODE, DEFAULT_M = 0, 1.35


class CosFace(_MarginHead):
    MODE, DEFAULT_M = 1, 0.35


class ArcFace(_MarginHead):
    MODE, DEFAULT_M = 2, 0.50


class CircleLoss:
    """Callable with the reference's signature plus an explicit backward (the reference relies on TF autodiff)."""

    def __init__(self, bufs, key="circle"):
        self.bufs, self.key = bufs, key

    def forward(self, y_true, y_pred, gamma=256, margin=0.25):
        B, A = y_pred.shape
        self.y, self.s, self.gamma, self.margin = y_true, y_pred, float(gamma), float(margin)
        self.loss = self.bufs.get(self.key + "/loss", (B,))
        _call("sar_circle_loss_fwd_f32", y_true, y_pred, self.loss, B, A, self.gamma, self.margin)
        return self.loss

    def backward(self, dloss):
        B, A = self.s.shape
        ds = self.bufs.get(self.key + "/ds", (B, A))
        _call("sar_circle_loss_bwd_f32", self.y, self.s, dloss, ds, B, A, self.gamma, self.margin)
        return ds


def circle_loss(y_true, y_pred, gamma=256, margin=0.25):
    """losses.py:157-172 (forward only; SAR_Net uses CircleLoss for the train step)."""
    import torch
    B, A = y_pred.shape
    loss = torch.empty(B, dtype=torch.float32, device=y_pred.device)
    _call("sar_circle_loss_fwd_f32", y_true, y_pred, loss, B, A, float(gamma), float(margin))
    return loss


class CategoricalCrossentropy:
    """Keras categorical_crossentropy on probabilities (model.py:350,354) with explicit backward."""

    def __init__(self, bufs, key):
        self.bufs, self.key = bufs, key

    def forward(self, y_true, p):
        B, A = p.shape
        self.y, self.p = y_true, p
        self.loss = self.bufs.get(self.key + "/loss", (B,))
        _call("sar_cce_fwd_f32", y_true, p, self.loss, B, A)
        return self.loss

    def backward(self, dloss):
        B, A = self.p.shape
        dp = self.bufs.get(self.key + "/dp", (B, A))
        _call("sar_cce_bwd_f32", self.y, self.p, dloss, dp, B, A)
        return dp
