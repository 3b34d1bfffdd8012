"""Layer-by-layer deviation of the bf16 trunk from the fp32-storage trunk (same weights, same batch) on the device:
activation relative L2 error per block, gradient cosine / relative error per parameter tensor in backward order.
Usage: python tools/bf16_diag.py [B] [T]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aesrc2020_amd import _lib
from aesrc2020_amd.model import SAR_Net
from aesrc2020_amd.utils import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
kw = dict(res_type="res34", res_filters=32, hidden_dim=256, mto="gvlad", vlad_clusters=32, ghost_clusters=2, disc_enable=True,
          ctc_enable=False, metric_loss="arcface", margin=0.3)
shape = [T, 80, 1]
models = {}
for dt, math in (("fp32", "fp32"), ("bf16", "bf16x3")):
    _lib.set_trunk_dtype(dt); _lib.set_conv_math(math)
    models[dt] = SAR_Net(shape, seed=0, use_graph=False, **kw)[0]
m32, m16 = models["fp32"], models["bf16"]
m16.set_weights_dict({k: v for k, v in m32.store.state_dict().items()})
xi, yi = synthetic_batch(shape, B, False, True, seed=5)
for m in (m32, m16):
    m.forward_backward_only(xi, yi)
torch.cuda.synchronize()


def rel(a, b):
    a, b = a.float().double(), b.float().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


print("forward: relative L2 error of the bf16 trunk vs fp32 storage")
print("  pooled %.3e" % rel(m16.trunk.blocks[0]["x"], m32.trunk.blocks[0]["x"]))
for n, (b16, b32) in enumerate(zip(m16.trunk.blocks, m32.trunk.blocks)):
    print("  block %2d stage %d: c1 %.3e  x_in %.3e" % (n, b32["stage"], rel(b16["c1"], b32["c1"]), rel(b16["x"], b32["x"])))
print("  pre_final %.3e  trunk out %.3e" % (rel(m16.trunk.pre_final, m32.trunk.pre_final), rel(m16.trunk.output, m32.trunk.output)))
for k in m32.out:
    print("  out %-10s max abs dev %.3e" % (k, float((m16.out[k] - m32.out[k]).abs().max())))
print("backward: per-parameter gradient, reverse creation order (cosine, relative L2 error, |g_ref|)")
for r16 in reversed(m16.store.refs):
    r32 = m32.store.by_name[r16.name]
    a, b = r16.g.double().ravel(), r32.g.double().ravel()
    cos = float(a @ b / (a.norm() * b.norm() + 1e-300))
    print("  %-34s cos %.5f rel %.3e  norm %.3e" % (r16.name, cos, rel(r16.g, r32.g), float(b.norm())))
