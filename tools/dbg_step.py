import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aesrc2020_amd.model import SAR_Net
from aesrc2020_amd.utils import synthetic_batch
import contextlib, io
B = int(sys.argv[1]); T = int(sys.argv[2])
kw = dict(res_type="res34", res_filters=32, hidden_dim=256, mto="gvlad", vlad_clusters=32, ghost_clusters=2, disc_enable=True, ctc_enable=False, metric_loss="arcface", margin=0.3)
with contextlib.redirect_stdout(io.StringIO()):
    model, tm = SAR_Net([T,80,1], use_graph=False, **kw)
xi, yi = synthetic_batch([T,80,1], B, False, True)
model._upload(xi, yi)
def t(name, fn):
    torch.cuda.synchronize(); t0=time.time(); fn(); torch.cuda.synchronize(); print("%-12s %.1f ms" % (name, (time.time()-t0)*1e3), flush=True)
for it in range(2):
    t("zero", model.store.zero_grads)
    t("forward", model._forward)
    t("losses", model._losses_forward)
    t("backward", model._backward)
    t("adam", model.store.adam_step)
print(model._logs(), flush=True)
