"""NetVLAD / GhostVLAD pooling layer, MI355X-native (mirrors /root/reference/VLAD.py:6-49).

`VladPooling(mode, k_centers, g_centers=0, name=...)` keeps the reference constructor; the weight is
`centers (K+G, D)` with an orthogonal initialiser (VLAD.py:17-19).  `layer([feat, cluster_score])` runs the fused
HIP kernel (csrc/vlad.hip) -- the reference's B x L x (K+G) x D broadcast tensor (VLAD.py:40-41) is never built."""
import numpy as np

from . import _lib
from .engine import orthogonal
from .layers import Layer

_call = _lib.call


class VladPooling(Layer):
    def __init__(self, mode, k_centers, g_centers=0, name=None, feat_dim=None, **kwargs):
        super().__init__(name or (mode + "_pool"))
        if mode not in ("vlad", "gvlad"):
            raise ValueError("mode must be 'vlad' or 'gvlad'")
        self.mode, self.k_centers = mode, int(k_centers)
        self.g_centers = int(g_centers) if mode == "gvlad" else 0
        self.cluster = None
        if feat_dim is not None:
            self.build(feat_dim)

    def build(self, feat_dim):
        kg = self.k_centers + self.g_centers
        init = orthogonal(self.store.rng, (kg, feat_dim)) if kg <= feat_dim else orthogonal(self.store.rng, (feat_dim, kg)).T
        self.cluster = self.store.add(self.name + "/centers", (kg, feat_dim), init)
        self.D = feat_dim
        self.output_shape = (None, self.k_centers * feat_dim)

    def compute_output_shape(self, input_shape):
        return (input_shape[0][0], self.k_centers * input_shape[0][-1])

    def __call__(self, x):
        return self.forward(*x)

    def forward(self, feat, cluster_score):
        """feat (B, [1,] L, D), cluster_score (B, [1,] L, K+G) -> (B, K*D)."""
        D, K, G = self.D, self.k_centers, self.g_centers
        B = feat.shape[0]; L = feat.numel() // (B * D)
        self.feat, self.B, self.L = feat, B, L
        self.out = self.buf("out", (B, K * D)); self.A = self.buf("A", (B, L, K + G)); self.inv = self.buf("inv", (B, K))
        _call("sar_vlad_fwd_f32", feat, cluster_score, self.cluster.w, self.out, self.A, self.inv, B, L, D, K, G)
        return self.out

    def backward(self, dout, dfeat_out=None, accumulate_dfeat=False):
        D, K, G, B, L = self.D, self.k_centers, self.g_centers, self.B, self.L
        dfeat = dfeat_out if dfeat_out is not None else self.buf("dfeat", (B, L, D))
        dscore = self.buf("dscore", (B, L, K + G))
        _call("sar_vlad_bwd_f32", dout, self.feat, self.A, self.cluster.w, self.out, self.inv, dfeat, dscore, self.cluster.g,
              B, L, D, K, G, 1 if accumulate_dfeat else 0)
        return dfeat, dscore
