"""Shape / option sweep of the whole train step against the float64 oracle: odd frame counts (TF-SAME padding on odd sizes at
every stride-2 layer), batch sizes down to 1, both trunks, several widths, every pooling mode and every head, with and without
the CTC branch -- forward outputs, every gradient, and one optimizer step.  Seeds are fixed; sizes stay small enough for the
oracle to finish in about a second per case."""
import numpy as np
import pytest
import torch

from oracle import model_torch as MT
from oracle import ops_np as O

pytestmark = pytest.mark.gpu

SWEEP = [
    dict(input_shape=[33, 80, 1], res_type="res18", res_filters=16, hidden_dim=16, mto="avg", B=2),
    dict(input_shape=[65, 80, 1], res_type="res18", res_filters=32, hidden_dim=32, mto="vlad", vlad_clusters=3, disc_enable=True,
         metric_loss="cosface", B=3),
    dict(input_shape=[97, 80, 1], res_type="res34", res_filters=16, hidden_dim=32, mto="gvlad", vlad_clusters=5, ghost_clusters=3,
         disc_enable=True, metric_loss="sphereface", margin=1.35, B=2),
    dict(input_shape=[130, 80, 1], res_type="res18", res_filters=16, hidden_dim=64, mto="bigru", disc_enable=True,
         metric_loss="softmax", B=5),
    dict(input_shape=[81, 80, 1], res_type="res18", res_filters=32, hidden_dim=32, mto="gvlad", vlad_clusters=8, ghost_clusters=2,
         disc_enable=True, ctc_enable=True, bpe_classes=17, max_ctc_len=5, metric_loss="circleloss", B=4),
    dict(input_shape=[64, 80, 1], res_type="res34", res_filters=32, hidden_dim=16, mto="avg", ctc_enable=True, ar_enable=True,
         bpe_classes=11, max_ctc_len=4, B=2),
    dict(input_shape=[49, 80, 1], res_type="res18", res_filters=16, hidden_dim=32, mto="gvlad", vlad_clusters=2, ghost_clusters=1,
         disc_enable=True, metric_loss="arcface", margin=0.5, accent_classes=5, B=7),
    dict(input_shape=[100, 80, 1], res_type="res18", res_filters=64, hidden_dim=32, mto="vlad", vlad_clusters=4, disc_enable=True,
         metric_loss="arcface", B=2),
    # other feature dimensions (the stage-0 map is 10 / 16 wide instead of 20: other patch geometries of every conv form)
    dict(input_shape=[72, 40, 1], res_type="res18", res_filters=32, hidden_dim=32, mto="gvlad", vlad_clusters=4, ghost_clusters=1, disc_enable=True,
         metric_loss="cosface", B=3),
    dict(input_shape=[56, 64, 1], res_type="res34", res_filters=32, hidden_dim=32, mto="avg", disc_enable=True, metric_loss="softmax", B=2),
    # hidden_dim past the persistent GRU kernels' 256 units (round 4: one GEMM + one gate launch per time step), two Bi-GRUs + Bi-GRU pooling
    dict(input_shape=[64, 80, 1], res_type="res18", res_filters=16, hidden_dim=320, mto="bigru", disc_enable=True, ctc_enable=True,
         bpe_classes=13, max_ctc_len=4, metric_loss="cosface", B=3),
]


# round 5: the widths the SHIPPED kernels take (res_filters 32: csrc/conv_bf16.hip in split precision) at the odd sizes --
# TF-SAME padding on odd extents at every stride-2 layer (resnet.py:113-122), batches of 1 / 2 / 5 / 7, 40- and 64-bin features
SWEEP += [
    dict(input_shape=[33, 80, 1], res_type="res18", res_filters=32, hidden_dim=32, mto="avg", B=1),
    dict(input_shape=[65, 80, 1], res_type="res34", res_filters=32, hidden_dim=32, mto="gvlad", vlad_clusters=4, ghost_clusters=2, disc_enable=True,
         metric_loss="arcface", B=2),
    dict(input_shape=[81, 40, 1], res_type="res18", res_filters=32, hidden_dim=64, mto="vlad", vlad_clusters=5, disc_enable=True,
         metric_loss="circleloss", B=5),
    dict(input_shape=[97, 64, 1], res_type="res34", res_filters=32, hidden_dim=32, mto="bigru", disc_enable=True, metric_loss="cosface", B=7),
    dict(input_shape=[131, 80, 1], res_type="res18", res_filters=32, hidden_dim=256, mto="gvlad", vlad_clusters=8, ghost_clusters=2, disc_enable=True,
         ctc_enable=True, bpe_classes=29, max_ctc_len=6, metric_loss="circleloss", B=2),       # hidden_dim 256: the fp16 MFMA recurrence
    dict(input_shape=[131, 64, 1], res_type="res34", res_filters=32, hidden_dim=32, mto="avg", disc_enable=True, metric_loss="sphereface", margin=1.35, B=5),
    dict(input_shape=[97, 40, 1], res_type="res18", res_filters=32, hidden_dim=32, mto="gvlad", vlad_clusters=3, ghost_clusters=1, disc_enable=True,
         ctc_enable=True, bpe_classes=12, max_ctc_len=4, metric_loss="softmax", B=7),
    dict(input_shape=[33, 64, 1], res_type="res18", res_filters=64, hidden_dim=32, mto="avg", disc_enable=True, metric_loss="cosface", B=2),
    dict(input_shape=[65, 40, 1], res_type="res34", res_filters=32, hidden_dim=32, mto="vlad", vlad_clusters=2, disc_enable=True, metric_loss="arcface", B=1),
    dict(input_shape=[81, 80, 1], res_type="res34", res_filters=32, hidden_dim=64, mto="gvlad", vlad_clusters=8, ghost_clusters=2, disc_enable=True,
         ctc_enable=True, bpe_classes=40, max_ctc_len=5, metric_loss="arcface", B=5),
    # four accents, nine utterances: the one-launch heads (csrc/heads_fused.hip) in their four-class form, the last row block ragged
    dict(input_shape=[65, 80, 1], res_type="res18", res_filters=32, hidden_dim=64, mto="gvlad", vlad_clusters=4, ghost_clusters=1, disc_enable=True,
         metric_loss="cosface", accent_classes=4, B=9),
]

# every case in the two fp32-storage modes of round 1; every case whose width the shipped kernels take (res_filters 32 / 64) ALSO in the
# package default "x3" -- the mode that ships was the least swept
CASES = [(i, m) for i in range(len(SWEEP)) for m in ("fp32", "bf16x3_r1")] + \
        [(i, "x3") for i in range(len(SWEEP)) if SWEEP[i]["res_filters"] in (32, 64)]


# the shipped (split-precision) mode's gradient bounds (round 6; 3e-2 on the flat vector alone until then).  Measured over the sweep's 17 shapes in
# that mode (gpurun, r6): flat vector 9e-4 .. 1.24e-2 (the largest at 4 x 3 stage-3 pixels per utterance, B = 3), worst tensor -- max-abs error relative
# to the tensor's largest element -- 1e-3 .. 0.20: at these sizes (tensors of six rows, B <= 9) flipped ReLU / max-pool decisions move up to twenty
# tensors by percents, so the per-tensor view is a ceiling on the WORST tensor here (the "at most three tensors" rule of the exact-fp32 arm holds at
# the golden / whole-graph sizes: tests/test_gpu_golden.py, tests/test_gpu_model.py)
X3_FLAT, X3_WORST = 2e-2, 0.4


@pytest.mark.parametrize("i,mode_name", CASES, ids=["%d-%s" % c for c in CASES])
def test_config_sweep(i, mode_name):
    from tests._modes import mode
    with mode(mode_name):
        _config_sweep(i, mode_name)


def _config_sweep(i, conv_math):
    from aesrc2020_amd.model import SAR_Net
    from tests._modes import runs_shipped_kernels
    kw = dict(SWEEP[i]); B = kw.pop("B")
    cfg = MT.Cfg(**kw)
    ref = MT.TorchSARNet(cfg, seed=100 + i)
    model, _ = SAR_Net(seed=i, **kw)
    assert runs_shipped_kernels(model) == (conv_math == "x3")
    model.set_weights_dict({k: v.numpy() for k, v in ref.P.items()})
    assert set(r.name for r in model.store.refs) == set(ref.P)
    batch = MT.synthetic_batch(cfg, B, seed=500 + i)
    out, total, parts, G = ref.forward_backward(batch)
    p0 = {r.name: r.w.cpu().double().numpy() for r in model.store.refs}
    logs = model.train_on_batch(batch, batch)                  # eager step: forward + backward + L2 + Adam (+ unit_norm)
    for k in ("y_accent", "y_disc"):
        if k in out and k in model.out:
            d = np.abs(model.out[k].cpu().numpy() - out[k].detach().numpy()).max()
            assert d < 1e-3, (i, k, d)
    if cfg.ctc_enable:
        got = model.out["y_ctc_loss"].cpu().numpy(); want = out["y_ctc_loss"].detach().numpy()[:, 0]
        assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max(), (i, got, want)
    assert abs(logs["loss"] - float(total)) <= 1e-3 * abs(float(total)) + 1e-3, (i, logs["loss"], float(total))
    # Gradients (device grads exclude the L2 term; the Adam kernel adds it).  Two views:
    #  * the whole flat gradient: relative L2 error 2e-3 (exact-fp32 convs) / 1e-2 (bf16x3);
    #  * per tensor, max-abs relative to the tensor's max: 5e-3 -- except that these cases have tensors with as few as 6
    #    rows, where ONE ReLU / max-pool decision that differs from the float64 oracle (a pre-activation within rounding of
    #    zero) moves a weight or beta gradient by percents; at most three tensors may exceed the bound, by no more than 0.1
    #    (exact-fp32 mode) -- the split-precision mode is held to the flat-vector bound only.
    ref_g = {r.name: G[r.name].numpy() - (2 * MT.L2 * ref.P[r.name].numpy() if r.name in ref.reg else 0.0) for r in model.store.refs}
    num = sum(float(((r.g.cpu().double().numpy() - ref_g[r.name]) ** 2).sum()) for r in model.store.refs)
    den = sum(float((ref_g[r.name] ** 2).sum()) for r in model.store.refs)
    # ("x3": conv weight gradients, the stem's last product and the hidden_dim-256 recurrence are single-pass fp16 under the loss scale, and more
    # ReLU / max-pool decisions sit within the arithmetic's noise of zero: measured 1e-3 .. 8e-3, the whole-graph bound of tests/test_gpu_bf16_model.py)
    print("sweep %d %s: flat gradient rel. L2 %.3e" % (i, conv_math, (num / den) ** 0.5))
    # (B = 1: the 2-D BatchNorms AR_BN1 / AR_BN2 see a batch of one -- their output is beta, every gradient in front of them is zero up to
    # rounding, in the oracle and on the device: compared against the scale of the parameters' own gradients behind them, not against ~0)
    floor = 1e-12 * max(float(np.abs(v).max()) for v in ref_g.values()) ** 2 * len(ref_g)
    # (round 6, VERDICT r5 item 6: the shipped mode's flat bound is 2 x the worst measured -- 3e-2 until then, under which a backward regression of 3 x passed)
    assert num ** 0.5 <= {"fp32": 2e-3, "bf16x3_r1": 1e-2, "x3": X3_FLAT}[conv_math] * max(den, floor) ** 0.5, (i, conv_math, (num / max(den, floor)) ** 0.5)
    if conv_math in ("fp32", "x3"):
        # per tensor.  fp32: 5e-3, at most three tensors over it (ONE flipped ReLU / max-pool decision moves a six-row tensor's gradient by percents), none
        # by more than 0.1; the shipped mode: no tensor over X3_WORST (above)
        x3m = conv_math == "x3"
        tol, flip = (5e-3, 0.1) if not x3m else (X3_WORST, X3_WORST)
        med = float(np.median([np.abs(v).max() for v in ref_g.values()]))
        bad, worst = [], 0.0
        for r in model.store.refs:
            g = r.g.cpu().numpy(); gr = ref_g[r.name]
            if np.abs(gr).max() < 1e-9 * med:
                # (the shipped mode: a structurally zero bias gradient is a column sum of the fp16 copy of a gradient tensor whose exact sums cancel)
                if not np.abs(g).max() < (2e-2 if x3m else 1e-4) * med:
                    bad.append((r.name, "zero-grad tensor", 1.0))
                continue
            e = np.abs(g - gr).max() / max(np.abs(gr).max(), 1e-3 * med, 1e-30)
            worst = max(worst, float(e))
            if not e < tol:
                bad.append((r.name, float(e), float(np.abs(gr).max())))
        print("sweep %d %s: worst tensor max-abs rel. %.3e, %d over %.0e" % (i, conv_math, worst, len(bad), tol))
        assert len(bad) <= (0 if x3m else 3) and all(b[1] != "zero-grad tensor" and b[1] < flip for b in bad), (i, med, bad[:10])
    # the update: Keras-Adam (float64 restatement) applied to the device gradients
    for r in model.store.refs:
        g = r.g.cpu().double().numpy() + 2 * r.l2 * p0[r.name]
        want, _, _ = O.adam_step(p0[r.name], g, np.zeros(r.shape), np.zeros(r.shape), 0, lr=cfg.lr)
        if r.constraint == "unit_norm":
            want = O.unit_norm(want)
        got = r.w.cpu().double().numpy()
        assert np.abs(got - want).max() <= 1e-4 * cfg.lr + 1e-5 * np.abs(want).max(), (i, r.name)
