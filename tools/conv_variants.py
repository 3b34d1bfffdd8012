"""Conv launch classes + step time of cfg2 under the library named by SAR_LIB_PATH (developer tool: phase-removal variants of
csrc/conv_bf16.hip linked into a full library tell where a conv launch's time goes; their results are wrong by construction).
    SAR_LIB_PATH=tools/libsar_<variant>.so python tools/conv_variants.py"""
import contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from aesrc2020_amd import _lib
from aesrc2020_amd.model import SAR_Net
from aesrc2020_amd.utils import synthetic_batch

_lib.set_trunk_dtype("bf16"); _lib.set_conv_math("bf16x3"); _lib.set_gemm_math("bf16")
kw = bench.WORKLOADS["cfg2"]["kw"]
B, shape = 64, [400, 80, 1]
with contextlib.redirect_stdout(io.StringIO()):
    model, tm = SAR_Net(shape, lr=0.01, gpus=1, seed=0, use_graph=True, **kw)
xi, yi = synthetic_batch(shape, B, False, True, seed=1234)
d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in {**xi, **yi}.items()}
for _ in range(4 + (150 if os.environ.get("CV_STEP_ONLY") else 0)):
    tm.set_device_batch(d); tm.train_step_device()
torch.cuda.synchronize(); t0 = time.perf_counter()
NS = int(os.environ.get("CV_STEPS", 20))
for _ in range(NS):
    tm.set_device_batch(d); tm.train_step_device()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / NS * 1e3
if os.environ.get("CV_STEP_ONLY"):                                  # in-step A/B of two builds: several hundred steps, no isolated replays
    print("%-28s step %.4f ms (%d steps)" % (os.path.basename(os.environ.get("SAR_LIB_PATH", "default")), ms, NS)); sys.exit(0)
r = bench.conv_roofline(model, B)
print("%-28s step %.3f ms conv_total %.1f us | %s" % (os.path.basename(os.environ.get("SAR_LIB_PATH", "default")), ms, r["conv_total_us_per_step"],
      " ".join("%s=%.1f" % (k.replace("conv_bf16_kernel", "c").replace("wgrad", "w").replace("_bf16_kernel", "").replace(" ", ""), v["us_per_step"]) for k, v in r["conv_classes"].items())))
