#!/usr/bin/env python
"""Developer diagnostic: per-tensor gradient deviation from the float64 oracle of the whole-graph parity cases at res_filters 32 in several
arithmetic modes (tests/_modes.py) -- separates arithmetic (a mode-specific deviation) from discrete ReLU / max-pool flips (every mode)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model_torch as MT
from tests._modes import mode
from tests.test_gpu_model import CASES, build_pair

cases = sys.argv[1:] or ["cfg2_gvlad_arcface", "cfg3_ctc_circle", "bigru_cosface_ctc", "cfg1_ar_only"]
for case in cases:
    for m in ("fp32", "bf16x3_r1", "x3", "x3trunk_f32gemm"):
        from aesrc2020_amd import _lib
        with mode("x3" if m.startswith("x3") else m):
            if m == "x3trunk_f32gemm":
                _lib.set_gemm_math("fp32")
            kw = dict(CASES[case], res_filters=32)
            cfg, ref, model, batch = build_pair(kw)
            out, total, parts, G = ref.forward_backward(batch)
            model.forward_backward_only(batch, batch); torch.cuda.synchronize()
            ref_g = {r.name: G[r.name].numpy() - (2 * MT.L2 * ref.P[r.name].numpy() if r.name in ref.reg else 0.0) for r in model.store.refs}
            med = float(np.median([np.abs(v).max() for v in ref_g.values()]))
            rows = []
            for r in model.store.refs:
                g = r.g.cpu().numpy(); gr = ref_g[r.name]
                if np.abs(gr).max() < 1e-9 * med:
                    continue
                e = np.abs(g - gr).max() / max(np.abs(gr).max(), 1e-3 * med)
                c = float((g.ravel() @ gr.ravel()) / (np.linalg.norm(g) * np.linalg.norm(gr) + 1e-300))
                rows.append((float(e), c, r.name, float(np.abs(gr).max())))
            rows.sort(reverse=True)
            dacc = float(np.abs(model.out["y_accent"].cpu().numpy() - out["y_accent"].detach().numpy()).max())
            print("%s %-16s accent %.1e | worst: %s" % (case, m, dacc, "; ".join("%s %.1e cos %.5f" % (n, e, c) for e, c, n, _ in rows[:4])), flush=True)
