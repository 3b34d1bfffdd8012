#!/usr/bin/env python
"""In-kernel phase timing of the VLAD pooling kernels (developer tool):

    hipcc ... -DSAR_TIMING -shared aesrc2020_amd/csrc/vlad.hip -o tools/libsarnet_timing_vlad.so   (make -C aesrc2020_amd/csrc timing)
    python tools/vlad_timing.py [B L D K G]

Thread 0 of every block stamps the shader clock at phase boundaries; prints the distribution over blocks."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libsarnet_timing_vlad.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.sar_vlad_fwd_f32.argtypes = [vp] * 6 + [ci] * 5 + [vp]
lib.sar_dbg_set_timing_buffer_vlad.argtypes = [vp]
B, L, D, K, G = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (64, 39, 256, 32, 2)
feat = torch.randn(B, L, D, device="cuda"); score = torch.randn(B, L, K + G, device="cuda"); cen = torch.randn(K + G, D, device="cuda")
out = torch.zeros(B, K * D, device="cuda"); A = torch.zeros(B, L, K + G, device="cuda"); inv = torch.zeros(B, K, device="cuda")
st = torch.cuda.current_stream().cuda_stream
fn = lambda: lib.sar_vlad_fwd_f32(feat.data_ptr(), score.data_ptr(), cen.data_ptr(), out.data_ptr(), A.data_ptr(), inv.data_ptr(), B, L, D, K, G, st)
tb = torch.zeros(1 << 12, 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    lib.sar_dbg_set_timing_buffer_vlad(None); torch.cuda.synchronize(); assert fn() == 0
torch.cuda.synchronize()
lib.sar_dbg_set_timing_buffer_vlad(tb.data_ptr()); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); assert fn() == 0; e1.record(); torch.cuda.synchronize()
t = tb.cpu().numpy().astype(np.int64); t = t[t[:, 0] != 0]
rt0 = t[:, 0].min()
print("vlad fwd B=%d L=%d D=%d K=%d G=%d: blocks %d, event %.1f us; block starts max %.1f us, ends p50 %.1f max %.1f us"
      % (B, L, D, K, G, len(t), e0.elapsed_time(e1) * 1e3, (t[:, 0] - rt0).max() / 100.0, np.percentile(t[:, 15] - rt0, 50) / 100.0, (t[:, 15] - rt0).max() / 100.0))
for name, (a, b) in {"zero + assignments + feature staging": (1, 2), "assignments out + column sums": (2, 3), "MFMA loop": (3, 4), "residuals + row norms": (4, 5),
                     "barrier": (5, 6), "normalise + store": (6, 7), "total": (1, 7)}.items():
    v = t[:, b] - t[:, a]
    print("   %-40s cycles: p10 %7d  p50 %7d  p90 %7d  max %7d" % (name, np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
