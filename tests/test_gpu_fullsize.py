"""Full-size (BASELINE.json shapes) GPU tests through size-independent properties -- the float64 oracle is too slow here:
 * duplicated half-batch: BatchNorm batch statistics of [X; X] equal those of X, so every per-utterance output of the
   B=64 run must equal the B=32 run of the half, and both halves must agree with each other;
 * CTC sum rule at B=64, L=39, V=1000: sum_k (p_k + 1e-7) * dL/dp_k = 0 for every frame (state posteriors sum to 1);
 * GhostVLAD rows are unit vectors; head probabilities sum to 1; everything finite;
 * cfg5 long-utterance stress (T=2000 -> L=189, GhostVLAD 64+2) runs one full step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG3 = dict(input_shape=[400, 80, 1], res_type="res34", res_filters=32, hidden_dim=256, mto="gvlad", vlad_clusters=8, ghost_clusters=2,
            disc_enable=True, ctc_enable=True, metric_loss="circleloss", margin=0.3, bpe_classes=1000, max_ctc_len=72)


def _batch(kw, B, seed):
    from aesrc2020_amd.utils import synthetic_batch
    x, y = synthetic_batch(kw["input_shape"], B, kw.get("ctc_enable", False), kw.get("disc_enable", False),
                           bpe_classes=kw.get("bpe_classes", 1000), max_ctc_len=kw.get("max_ctc_len", 72), seed=seed)
    return {**x, **y}


def test_cfg3_duplicated_half_batch_invariance():
    from aesrc2020_amd.model import SAR_Net
    half = _batch(CFG3, 32, 5)
    full = {k: np.concatenate([v, v]) for k, v in half.items()}
    m64, _ = SAR_Net(seed=0, use_graph=False, **CFG3)
    m32, _ = SAR_Net(seed=0, use_graph=False, **CFG3)
    m32.set_weights_dict(m64.get_weights_dict())
    m64.forward_backward_only(full, full); m32.forward_backward_only(half, half)
    for k in ("y_accent", "y_disc", "y_ctc_loss"):
        a = m64.out[k].cpu().numpy(); b = m32.out[k].cpu().numpy()
        assert np.all(np.isfinite(a))
        scale = np.abs(b).max() + 1e-6
        assert np.abs(a[:32] - a[32:]).max() <= 2e-4 * scale, k       # split-K atomics reorder fp32 sums, hence not bitwise
        assert np.abs(a[:32] - b).max() <= 2e-4 * scale, k
    assert (m64.out["y_ctc_loss"].cpu().numpy() > 0).all()
    # mean-loss gradients of the doubled batch equal those of the half (same per-sample gradients, batch-mean)
    g64 = m64.store.grads.cpu().numpy(); g32 = m32.store.grads.cpu().numpy()
    assert np.all(np.isfinite(g64))
    assert np.abs(g64 - g32).max() <= 2e-2 * np.abs(g32).max()      # different tilings / split-K orders at B=64 vs B=32
    pooled = m64.merge.pool.out.view(64, 8, 256)
    assert torch.allclose(pooled.norm(dim=2), torch.ones(64, 8, device="cuda", dtype=torch.float32), atol=1e-4)
    assert torch.allclose(m64.out["y_accent"].sum(1), torch.ones(64, device="cuda", dtype=torch.float32), atol=1e-5)


def test_ctc_sum_rule_full_size():
    from aesrc2020_amd import _lib as L
    B, Lq, V, U = 64, 39, 1000, 72
    g = torch.Generator(device="cuda").manual_seed(3)
    p = torch.softmax(torch.randn(B, Lq, V, device="cuda", generator=g, dtype=torch.float32) * 3, -1).contiguous()
    rng = np.random.default_rng(1)
    lab = np.full((B, U), 2.0, np.float32); ll = rng.integers(4, 17, B).astype(np.int32)
    for b in range(B):
        lab[b, :ll[b]] = rng.integers(0, V - 1, ll[b])
    il = np.full(B, Lq, np.int32); il[::7] = 30
    labd = torch.tensor(lab, device="cuda"); ild = torch.tensor(il, device="cuda"); lld = torch.tensor(ll, device="cuda")
    wsb = L.ctc_workspace_bytes(B, Lq, U); ws = torch.zeros(wsb // 4, device="cuda", dtype=torch.float32)
    loss = torch.zeros(B, device="cuda", dtype=torch.float32); dp = torch.zeros(B, Lq, V, device="cuda", dtype=torch.float32)
    L.call("sar_ctc_batch_cost_fwd_f32", p, labd, ild, lld, loss, ws, wsb, B, Lq, V, U)
    L.call("sar_ctc_batch_cost_bwd_f32", p, labd, ild, lld, loss, torch.ones(B, device="cuda", dtype=torch.float32), ws, wsb, dp, B, Lq, V, U)
    assert torch.isfinite(loss).all() and (loss > 0).all()
    s = ((p.double() + 1e-7) * dp.double()).sum(-1)                 # (B, L)
    assert s.abs().max().item() < 1e-3
    for b in range(0, B, 7):
        assert (dp[b, 30:] == 0).all()                              # frames beyond in_len get exactly zero gradient


CFG2 = dict(input_shape=[400, 80, 1], res_type="res34", res_filters=32, hidden_dim=256, mto="gvlad", vlad_clusters=32, ghost_clusters=2,
            disc_enable=True, ctc_enable=False, metric_loss="arcface", margin=0.3)
CFG5 = dict(input_shape=[2000, 80, 1], res_type="res34", res_filters=32, hidden_dim=256, mto="gvlad", vlad_clusters=64, ghost_clusters=2,
            disc_enable=True, ctc_enable=False, metric_loss="arcface", margin=0.3)


# the reference's shipped configuration (/root/reference/train.py:49-75): 1200-frame inputs (L = 114), B = 32, Bi-GRU pooling, softmax heads, CTC on
TRAIN_PY = dict(input_shape=[1200, 80, 1], res_type="res34", res_filters=32, hidden_dim=256, mto="bigru", disc_enable=True, ctc_enable=True,
                metric_loss="softmax", margin=0.3, bpe_classes=1000, max_ctc_len=72)


def _dup_half(kw, B, tol_out, tol_grad, seed=11, tol_halves=None):
    """B-utterance batch = two copies of a B/2 batch: per-utterance outputs of the full run equal the half run's, halves agree,
    batch-mean gradients agree."""
    from aesrc2020_amd.model import SAR_Net
    half = _batch(kw, B // 2, seed)
    full = {k: np.concatenate([v, v]) for k, v in half.items()}
    mf, _ = SAR_Net(seed=0, use_graph=False, **kw)
    mh, _ = SAR_Net(seed=0, use_graph=False, **kw)
    mh.set_weights_dict(mf.get_weights_dict())
    mf.forward_backward_only(full, full); mh.forward_backward_only(half, half)
    if "y_ctc_loss" in mf.out:                                      # per-utterance CTC losses: positive, the same in both halves and in the half run
        a = mf.out["y_ctc_loss"].cpu().numpy(); b = mh.out["y_ctc_loss"].cpu().numpy()
        assert np.all(np.isfinite(a)) and (a > 0).all()
        assert np.abs(a[:B // 2] - a[B // 2:]).max() <= (tol_out if tol_halves is None else tol_halves) * np.abs(b).max()
        assert np.abs(a[:B // 2] - b).max() <= tol_out * np.abs(b).max()
    for k in ("y_accent", "y_disc"):
        a = mf.out[k].cpu().numpy(); b = mh.out[k].cpu().numpy()
        if kw.get("metric_loss") == "circleloss" and k == "y_disc":  # cosine logits, not probabilities
            assert np.all(np.isfinite(a)) and np.abs(a).max() <= 1.0 + 1e-4
        else:
            assert np.all(np.isfinite(a)) and np.abs(a.sum(1) - 1).max() < 1e-4
        assert np.abs(a[:B // 2] - a[B // 2:]).max() <= (tol_out if tol_halves is None else tol_halves), k
        assert np.abs(a[:B // 2] - b).max() <= tol_out, k
    gf = mf.store.grads.cpu().numpy(); gh = mh.store.grads.cpu().numpy()
    assert np.all(np.isfinite(gf))
    assert np.linalg.norm(gf - gh) <= tol_grad * np.linalg.norm(gh)
    return mf


@pytest.mark.parametrize("B", [32, 64])
def test_cfg5_duplicated_half_batch_invariance(B):
    """BASELINE configs[4] shape (2000x80, GhostVLAD 64+2) at its batch sizes, fp32-class default mode."""
    m = _dup_half(CFG5, B, 2e-4, 2e-2)
    assert m.L == 189 and m.merge.pool.out.shape == (B, 64 * 256)


def _two_byte(mode, det=False):
    from aesrc2020_amd import _lib
    _lib.set_trunk_dtype(mode); _lib.set_gemm_math({"fp32": "fp32", "bf16x3": "bf16x3"}.get(mode, "bf16")); _lib.set_bn_deterministic(det)


# B=64 against B=32 with atomically accumulated statistic bins: the two runs see other atomic orders, stored 2-byte values flip last
# bits and the runs agree to the recipe's noise level only (tools/dup_half_stats.py, bf16, 36 fresh runs: accent 1.4e-2 .. 3.2e-2, disc
# 2.4e-2 .. 4.6e-2 -- a 6e-2 bound failed once in three suite runs -- gradient 0.106 .. 0.121; fp16 sits 4-8x below), while the two halves
# of ONE run (same bins) agree to 7e-5 at most.  With deterministic statistics (below) the runs agree to the two-halves bound.
NOISE = {"bf16": (1e-1, 2.5e-1), "fp16": (2.5e-2, 8e-2)}


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_cfg2_two_byte_mode_duplicated_half_batch_invariance(mode):
    """The bench's default workload (cfg2, B=64) in both 2-byte trunk modes (+ bf16-MFMA GEMMs / GRU)."""
    _two_byte(mode)
    try:
        m = _dup_half(CFG2, 64, NOISE[mode][0], NOISE[mode][1], tol_halves=1e-3)
        assert m.trunk.bf16 and m.trunk.half == (mode == "fp16")
    finally:
        _two_byte("fp32")


@pytest.mark.parametrize("case,B", [("cfg2", 64), ("cfg3", 64), ("train_py", 32), ("cfg5", 32)])
def test_split_precision_mode_duplicated_half_batch_invariance(case, B):
    """The bench's DEFAULT mode (round 4: --dtype bf16x3, split precision in the trunk and the Dense / GRU-projection GEMMs, fp16 GRU recurrence)
    at the full size of every configuration bench.py times in it: cfg2 (the headline), cfg3, the reference's shipped train.py configuration
    and cfg5 -- outputs of the doubled batch equal those of the half to the fp32-class bound (measured 2e-5; the statistic bins are float
    atomics, their order moves 2^-24 effects), gradients to 2e-2 (ReLU / max-pool decisions that flip)."""
    kw = {"cfg2": CFG2, "cfg3": CFG3, "train_py": TRAIN_PY, "cfg5": CFG5}[case]
    _two_byte("bf16x3")
    try:
        m = _dup_half(kw, B, 5e-4, 3e-2)
        assert m.trunk.x3 and m.trunk.bf16 and not m.trunk.half
        if case == "train_py":
            assert m.L == 114
    finally:
        _two_byte("fp32")


def test_cfg3_fp16_mode_duplicated_half_batch_invariance():
    """cfg3 at full size in the fp16 build bench.py also times it in (only the fp32-class mode had a full-size cfg3 test)."""
    _two_byte("fp16")
    try:
        m = _dup_half(CFG3, 64, NOISE["fp16"][0], NOISE["fp16"][1], tol_halves=1e-3)
        assert m.trunk.half
    finally:
        _two_byte("fp32")


@pytest.mark.parametrize("B", [32, 64])
def test_cfg5_two_byte_mode_duplicated_half_batch_invariance(B):
    """BASELINE configs[4] as bench.py runs it: 2000 x 80, GhostVLAD 64+2, fp16 trunk, at B = 64 (SURVEY's table) and 32."""
    _two_byte("fp16")
    try:
        m = _dup_half(CFG5, B, NOISE["fp16"][0], NOISE["fp16"][1], tol_halves=1e-3)
        assert m.trunk.half and m.L == 189 and m.merge.pool.out.shape == (B, 64 * 256)
    finally:
        _two_byte("fp32")


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_cfg2_deterministic_statistics_make_the_batch_cut_invisible(mode):
    """SAR_BN_DET=1 / _lib.set_bn_deterministic(True): the forward BatchNorm sums are accumulated as 64-bit fixed
    point -- integer adds, so neither the atomic order nor the way a batch is cut into pixel tiles can change them.  The same 64
    utterances then give the same outputs alone and duplicated into a batch of 128, to the two-halves bound (1e-3; measured 4e-7: the
    trunk and every sequence layer are BIT-identical between the two runs -- tools/dbg_det.py --, what is left are fp32 summation
    orders in the two 2-D BatchNorms and split-K GEMMs behind the pooling), and two runs of one batch are bit-identical in the trunk.
    (128 against 64, not 64 against 32: the bf16-MFMA GEMM path takes problems from 64 rows up, so at B = 32 the 8192 -> 256 embedding
    layer runs on the fp32 kernel -- other arithmetic, 7e-3 on its output, nothing to do with statistics.)"""
    from aesrc2020_amd.model import SAR_Net
    _two_byte(mode, det=True)
    try:
        m = _dup_half(CFG2, 128, 1e-3, NOISE[mode][1], tol_halves=1e-3)
        assert m.trunk.det
        b = _batch(CFG2, 64, 5)
        m.forward_backward_only(b, b)
        first = {k: m.out[k].clone() for k in ("y_accent", "y_disc")}
        trunk_first = m.trunk.output.clone()
        m.forward_backward_only(b, b)
        assert torch.equal(trunk_first, m.trunk.output)                      # the whole trunk forward, bit for bit
        for k, v in first.items():
            assert (m.out[k] - v).abs().max().item() < 2e-4, k          # behind the trunk: split-K / LayerNorm-gradient fp32 atomics remain (measured 3e-5)
    finally:
        _two_byte("fp32")


def test_cfg5_long_utterance_stress():
    from aesrc2020_amd.model import SAR_Net
    kw = dict(input_shape=[2000, 80, 1], res_type="res34", res_filters=32, hidden_dim=256, mto="gvlad", vlad_clusters=64, ghost_clusters=2,
              disc_enable=True, ctc_enable=False, metric_loss="arcface", margin=0.3)
    m, _ = SAR_Net(seed=0, **kw)
    assert m.L == 189
    b = _batch(kw, 8, 9)
    logs = [m.train_on_batch(b, b) for _ in range(3)]
    assert all(np.isfinite(l["loss"]) for l in logs) and m.graph is not None
    assert m.merge.pool.out.shape == (8, 64 * 256)
