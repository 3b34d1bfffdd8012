import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from aesrc2020_amd import _lib
from aesrc2020_amd.model import SAR_Net
import test_gpu_fullsize as TF
mode = sys.argv[1] if len(sys.argv) > 1 else "fp16"
_lib.set_trunk_dtype(mode); _lib.set_gemm_math(os.environ.get("GM", "bf16")); _lib.set_bn_deterministic(True)
kw = TF.CFG2; B = 64
half = TF._batch(kw, B // 2, 11)
full = {k: np.concatenate([v, v]) for k, v in half.items()}
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    mf, _ = SAR_Net(seed=0, use_graph=False, **kw)
    mh, _ = SAR_Net(seed=0, use_graph=False, **kw)
mh.set_weights_dict(mf.get_weights_dict())
mf.forward_backward_only(full, full); mh.forward_backward_only(half, half)
tf, th = mf.trunk, mh.trunk
print("pooled equal:", torch.equal(tf.pooled16[:32], th.pooled16), torch.equal(tf.pooled16[:32], tf.pooled16[32:]))
for n, (bf, bh) in enumerate(zip(tf.blocks, th.blocks)):
    e1 = torch.equal(bf["c1"][:32], bh["c1"]); 
    print("block", n, "c1 equal", e1, "| max diff", float((bf["c1"][:32].float() - bh["c1"].float()).abs().max()))
    if not e1: break
print("trunk out equal:", torch.equal(tf.output[:32], th.output), float((tf.output[:32] - th.output).abs().max()))
for lay_f, lay_h in zip(mf.store.layers, mh.store.layers):
    of, oh = getattr(lay_f, "output", None), getattr(lay_h, "output", None)
    if isinstance(of, torch.Tensor) and isinstance(oh, torch.Tensor) and of.shape[0] == 2 * oh.shape[0]:
        print("%-28s max |full[:half] - half| = %.3e   (max |.| %.3e)" % (lay_f.name, float((of[:oh.shape[0]].float() - oh.float()).abs().max()), float(oh.float().abs().max())))
for k in ("y_accent", "y_disc"):
    print(k, float((mf.out[k][:32] - mh.out[k]).abs().max()))
