"""Spread of the duplicated-half-batch statistics of tests/test_gpu_fullsize.py in the bf16 mode (developer tool): N fresh runs, prints the
three numbers the test bounds (6e-2, 6e-2, 2.5e-1), to tell a tolerance at the edge from a rare wrong result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aesrc2020_amd import _lib
from aesrc2020_amd.model import SAR_Net
from tests.test_gpu_fullsize import CFG2, CFG5, _batch
_lib.set_trunk_dtype("bf16"); _lib.set_gemm_math("bf16")
CFG, B = (CFG5, 32) if (len(sys.argv) > 2 and sys.argv[2] == "cfg5") else (CFG2, 64)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    half = _batch(CFG, B // 2, 11)
    full = {k: np.concatenate([v, v]) for k, v in half.items()}
    mf, _ = SAR_Net(seed=0, use_graph=False, **CFG); mh, _ = SAR_Net(seed=0, use_graph=False, **CFG)
    mh.set_weights_dict(mf.get_weights_dict())
    mf.forward_backward_only(full, full); mh.forward_backward_only(half, half)
    o = []
    for k in ("y_accent", "y_disc"):
        a = mf.out[k].cpu().numpy(); b = mh.out[k].cpu().numpy()
        o += [np.abs(a[:B // 2] - a[B // 2:]).max(), np.abs(a[:B // 2] - b).max()]
    gf = mf.store.grads.cpu().numpy(); gh = mh.store.grads.cpu().numpy()
    print("run %d: accent halves %.3e vs-half %.3e | disc halves %.3e vs-half %.3e | grad rel %.3e" % (it, *o, np.linalg.norm(gf - gh) / np.linalg.norm(gh)), flush=True)
    del mf, mh
