"""Developer tool: run one bench workload N times in fresh models and report non-finite values in the step's state (bisecting an intermittent
NaN with the SAR_* switches in the environment).   python tools/nan_hunt.py cfg3 12"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aesrc2020_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
args = argparse.Namespace(gpus=1, steps=int(os.environ.get("NH_STEPS", 60)), warmup=5, settle=0, workload=name, dtype=os.environ.get("NH_DTYPE", "fp16"),
                          batch=None, frames=None, no_graph=False, no_cpu_baseline=True, no_roofline=True, no_also=True)
trunk, conv, gemm, _ = bench.DTYPES[args.dtype]
_lib.set_trunk_dtype(trunk); _lib.set_conv_math(conv); _lib.set_gemm_math(gemm)
bad = 0
for i in range(N):
    try:
        res, model = bench.run_workload(name, args, 0, 1, False)
        st = model.store
        nf = {k: int((~torch.isfinite(getattr(st, k))).sum().item()) for k in ("params", "grads", "m", "v")}
        print(i, "ok", res["final_loss"], float(st.l2_loss.item()), nf)
    except AssertionError as e:
        bad += 1
        print(i, "FAIL", str(e)[:400])
    torch.cuda.synchronize()
print("failures: %d of %d" % (bad, N))
