"""Timing of the bf16-MFMA GRU recurrence kernels (developer tool): python tools/gru_bench.py <lib.so> [B L]
Loads ONE shared library built from csrc/gru_mfma.hip alone (variants with phases removed tell where a step's time goes)."""
import ctypes
import os
import sys
import torch

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L = int(sys.argv[3]) if len(sys.argv) > 3 else 39
H = 256
z = lambda *s: torch.randn(*s, device="cuda") * 0.1
xa = [z(B, L, 3 * H), z(B, L, 3 * H)]
U = [z(H, 3 * H), z(H, 3 * H)]
bh = [z(3 * H), z(3 * H)]
pack = torch.empty(2 * 2 * 3 * H * H, dtype=torch.bfloat16, device="cuda")
out, gates, hprev = z(B, L, 2 * H), z(2, B, L, 4 * H), z(2, B, L, H)
dout, dai, dah = z(B, L, 2 * H), z(2, B, L, 3 * H), z(2, B, L, 3 * H)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def fwd():
    rc = lib.sar_bigru_fwd_bf16(P(xa[0]), P(xa[1]), P(pack), P(bh[0]), P(bh[1]), P(out), P(gates), P(hprev), B, L, H, st())
    assert rc == 0, rc


def bwd():
    rc = lib.sar_bigru_bwd_bf16(P(dout), P(gates), P(hprev), P(pack), P(dai), P(dah), B, L, H, st())
    assert rc == 0, rc


assert lib.sar_bigru_pack_bf16(P(U[0]), P(U[1]), P(pack), H, st()) == 0
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("%-28s %s B=%d L=%d rows=%s: %7.2f us = %.2f us per step" % (os.path.basename(sys.argv[1]), name, B, L, os.environ.get("SAR_GRU_ROWS", "auto"), us, us / L))
