"""The oracle (CPU, runs everywhere) and the HIP kernels (-m gpu) against the upstream-held constants of
tests/golden/upstream_kat.py: TensorFlow's ctc_loss_op_test.py testBasic (loss + gradient), Keras' backend_test.py test_ctc
(the K.ctc_batch_cost wrapper the reference calls, model.py:62-64) and the Keras categorical_crossentropy docstring example
(model.py:350).  These are the only numbers in the suite that neither side of the parity tests produced."""
import numpy as np
import pytest
import torch

from oracle import ops_np as O
from tests.golden import upstream_kat as K


def _batch():
    """The two sequences as the (2, 5, 6) batch Keras' test_ctc builds: labels padded to width 5."""
    p = np.stack(K.CTC_PROBS)
    lab = np.full((2, 5), -1.0); lens = []
    for b, t in enumerate(K.CTC_TARGETS):
        lab[b, :len(t)] = t; lens.append(len(t))
    return p, np.maximum(lab, 0.0), np.asarray([5, 5]), np.asarray(lens)


def test_oracle_ctc_reproduces_the_tensorflow_and_keras_constants():
    p, lab, il, ll = _batch()
    assert np.abs(p.sum(-1) - 1).max() < 2e-6                      # the published matrices are probability rows
    loss, grad = O.ctc_batch_cost_fwd_bwd(p, lab, il, ll)
    np.testing.assert_allclose(loss[:, 0], K.CTC_LOSS, rtol=0, atol=K.KERAS_CTC_ATOL)      # Keras' own tolerance for K.ctc_batch_cost
    # TF publishes d loss / d logits; the oracle returns d loss / d y_pred of K.ctc_batch_cost, logits = log(y_pred + 1e-7)
    for b in range(2):
        dlogits = (p[b] + 1e-7) * grad[b]
        np.testing.assert_allclose(dlogits, K.CTC_GRAD_LOGITS[b], rtol=0, atol=2e-6)        # 6 published digits


def test_oracle_categorical_crossentropy_reproduces_the_keras_docstring():
    loss, _ = O.categorical_crossentropy_fwd_bwd(K.CCE_Y_TRUE, K.CCE_Y_PRED)
    np.testing.assert_allclose(np.ravel(loss), K.CCE_LOSS, rtol=0, atol=K.CCE_ATOL)
    assert abs(float(np.mean(loss)) - K.CCE_MEAN) < K.CCE_ATOL


@pytest.mark.gpu
def test_hip_ctc_reproduces_the_tensorflow_and_keras_constants():
    from aesrc2020_amd import _lib as L
    p, lab, il, ll = _batch()
    B, Lq, V, maxU = 2, 5, 6, 5
    dev = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
    pd, labd, ild, lld = dev(p), dev(lab), dev(il, torch.int32), dev(ll, torch.int32)
    wsb = L.ctc_workspace_bytes(B, Lq, maxU); ws = torch.zeros(wsb // 4, device="cuda")
    loss = torch.zeros(B, device="cuda")
    L.call("sar_ctc_batch_cost_fwd_f32", pd, labd, ild, lld, loss, ws, wsb, B, Lq, V, maxU)
    np.testing.assert_allclose(loss.cpu().numpy(), K.CTC_LOSS, rtol=0, atol=K.KERAS_CTC_ATOL)
    dp = torch.zeros(B, Lq, V, device="cuda")
    L.call("sar_ctc_batch_cost_bwd_f32", pd, labd, ild, lld, loss, dev(np.ones(B)), ws, wsb, dp, B, Lq, V, maxU)
    got = dp.cpu().numpy().astype(np.float64)
    for b in range(2):
        np.testing.assert_allclose((p[b] + 1e-7) * got[b], K.CTC_GRAD_LOGITS[b], rtol=0, atol=5e-6)


@pytest.mark.gpu
def test_hip_categorical_crossentropy_reproduces_the_keras_docstring():
    from aesrc2020_amd import _lib as L
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    loss = torch.zeros(2, device="cuda")
    L.call("sar_cce_fwd_f32", dev(K.CCE_Y_TRUE), dev(K.CCE_Y_PRED), loss, 2, 3)
    got = loss.cpu().numpy()
    np.testing.assert_allclose(got, K.CCE_LOSS, rtol=0, atol=K.CCE_ATOL)
    assert abs(float(got.mean()) - K.CCE_MEAN) < K.CCE_ATOL


# ---------------------------------------------------------------------------------------------------------------------------------------
# Closed forms of the upstream semantics the reference delegates to where no published constant exists: inputs chosen so
# that the candidate readings of the semantics give DIFFERENT numbers, expectations written out by hand below (no oracle function is called
# to produce them).  Each is held against the oracle (CPU) and against the HIP kernels (-m gpu).
#   LayerNormalization (model.py:4,32-33): eps = K.epsilon()**2 = 1e-14.  A row +-d with d = 1e-6 has mean 0 and variance d^2 = 1e-12:
#     y = +-d / sqrt(d^2 + eps) * gamma + beta = +-0.99504 (eps 1e-14), +-3.2e-3 (eps 1e-7), +-3.2e-5 (eps 1e-3).  A constant row gives beta.
#   CuDNNGRU (model.py:44-50), one step from h0 = 0: reset-AFTER applies r to (h U_h + b_hh) = b_hh, reset-before would apply it to h
#     (= 0) and add b_hh unscaled: hh = tanh(x W_h + b_ih + r * b_hh) vs tanh(x W_h + b_ih + b_hh); h1 = (1 - z) * hh.
#   BatchNormalization moving variance after one step from 1.0 (momentum 0.99) on a 2-row batch (a, b): batch variance v = ((a-b)/2)^2;
#     4-D fused path: Bessel-corrected, 0.99 + 0.01 * 2 v; 2-D path (Keras 2.2.x): v * n / (n - (1 + eps)) = 0.99 + 0.01 * v * 2 / 0.999;
#     a biased update would give 0.99 + 0.01 v.
LN_D, LN_EPS = 1e-6, 1e-14
LN_Y = LN_D / np.sqrt(LN_D ** 2 + LN_EPS)                     # 0.995037...


def _ln_case():
    x = np.zeros((3, 8)); x[0] = 0.75                         # constant row
    x[1] = LN_D * np.array([1, -1, 1, -1, 1, -1, 1, -1.0]); x[2] = -x[1]
    gamma = np.linspace(0.5, 2.0, 8); beta = np.linspace(-1.0, 1.0, 8)
    sgn = np.sign(x[1] / LN_D)
    want = np.stack([beta, sgn * LN_Y * gamma + beta, -sgn * LN_Y * gamma + beta])
    return x, gamma, beta, want


def _gru_case(H=4, I=3):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 1, I)); W = rng.standard_normal((I, 3 * H)) * 0.5; U = rng.standard_normal((H, 3 * H)) * 0.5
    bias = rng.standard_normal(6 * H) * 0.7                  # [b_iz, b_ir, b_ih, b_hz, b_hr, b_hh], all non-zero
    xa = x[:, 0] @ W + bias[:3 * H]
    sig = lambda a: 1.0 / (1.0 + np.exp(-a))
    z = sig(xa[:, :H] + bias[3 * H:4 * H]); r = sig(xa[:, H:2 * H] + bias[4 * H:5 * H])
    h_after = (1 - z) * np.tanh(xa[:, 2 * H:] + r * bias[5 * H:])
    h_before = (1 - z) * np.tanh(xa[:, 2 * H:] + bias[5 * H:])
    assert np.abs(h_after - h_before).max() > 0.05           # the case tells the two apart
    return x, W, U, bias, h_after


def _bn_case():
    a = np.array([1.0, -2.0, 0.5, 3.0]); b = np.array([3.0, 0.0, 0.5, -1.0])
    v = ((a - b) / 2) ** 2
    return np.stack([a, b]), 0.99 + 0.01 * 2 * v, 0.99 + 0.01 * v * 2 / (2 - (1 + 1e-3)), 0.01 * (a + b) / 2


def test_oracle_layernorm_epsilon_closed_form():
    x, gamma, beta, want = _ln_case()
    got, _ = O.layernorm_fwd(x, gamma, beta)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    from oracle import model_torch as MT
    got2 = MT.ln(torch.tensor(x), torch.tensor(gamma), torch.tensor(beta)).numpy()
    np.testing.assert_allclose(got2, want, rtol=0, atol=1e-9)


def test_oracle_gru_one_step_is_reset_after():
    x, W, U, bias, want = _gru_case()
    out, _ = O.gru_dir_fwd(x, W, U, bias)
    np.testing.assert_allclose(out[:, 0], want, rtol=0, atol=1e-12)
    from oracle import model_torch as MT
    out2 = MT.gru_dir(torch.tensor(x), torch.tensor(W), torch.tensor(U), torch.tensor(bias), False).numpy()
    np.testing.assert_allclose(out2[:, 0], want, rtol=0, atol=1e-12)


def test_oracle_batchnorm_moving_variance_closed_form():
    from oracle import model_torch as MT
    x, mv4, mv2, mm = _bn_case()
    for shape, want in (((2, 1, 1, 4), mv4), ((2, 4), mv2)):
        P = {"bn/gamma": torch.ones(4, dtype=torch.float64), "bn/beta": torch.zeros(4, dtype=torch.float64)}
        st = {"bn/moving_mean": np.zeros(4), "bn/moving_variance": np.ones(4)}
        MT.bn(torch.tensor(x.reshape(shape)), P, "bn", st, True)
        np.testing.assert_allclose(st["bn/moving_variance"], want, rtol=0, atol=1e-12)
        np.testing.assert_allclose(st["bn/moving_mean"], mm, rtol=0, atol=1e-12)


@pytest.mark.gpu
def test_hip_layernorm_epsilon_closed_form():
    from aesrc2020_amd import _lib as L
    x, gamma, beta, want = _ln_case()
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    y = torch.zeros(3, 8, device="cuda"); mean = torch.zeros(3, device="cuda"); rstd = torch.zeros(3, device="cuda")
    from aesrc2020_amd.layers import LN_EPS as PRODUCT_LN_EPS             # what the product's LayerNormalization layer passes
    assert PRODUCT_LN_EPS == LN_EPS
    L.call("sar_layernorm_fwd_f32", dev(x), dev(gamma), dev(beta), y, mean, rstd, 3, 8, PRODUCT_LN_EPS)
    got = y.cpu().numpy()
    np.testing.assert_allclose(got[0], beta, rtol=0, atol=1e-6)            # constant row: beta exactly
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)               # (fp32 input: d^2 and the sum of eight of them are rounded)


@pytest.mark.gpu
def test_hip_gru_one_step_is_reset_after():
    from aesrc2020_amd import _lib as L
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    for H, kern in ((4, "sar_bigru_fwd_f32"), (256, "sar_bigru_fwd_f32"), (256, "sar_bigru_fwd_bf16"), (256, "sar_bigru_fwd_f16")):
        x, W, U, bias, want = _gru_case(H=H, I=3)
        B = x.shape[0]
        xa = dev(x[:, 0] @ W + bias[:3 * H]).reshape(B, 1, 3 * H).contiguous()
        out = torch.zeros(B, 1, 2 * H, device="cuda"); gates = torch.zeros(2, B, 1, 4, H, device="cuda"); hprev = torch.zeros(2, B, 1, H, device="cuda")
        bh = dev(bias[3 * H:])
        if kern == "sar_bigru_fwd_f32":
            L.call(kern, xa, xa, dev(U), dev(U), bh, bh, out, gates, hprev, B, 1, H)
        else:
            sfx = kern[-4:].lstrip("_")
            pack = torch.zeros(2 * 2 * 3 * H * H, dtype=torch.float16 if sfx == "f16" else torch.bfloat16, device="cuda")
            L.call("sar_bigru_pack_" + sfx, dev(U), dev(U), pack, H)
            L.call(kern, xa, xa, pack, bh, bh, out, gates, hprev, B, 1, H)
        got = out.cpu().numpy()[:, 0]
        np.testing.assert_allclose(got[:, :H], want, rtol=0, atol=2e-6)   # forward direction; h0 = 0: the recurrent product is exactly zero
        np.testing.assert_allclose(got[:, H:], want, rtol=0, atol=2e-6)   # backward direction (same weights, one step): the same


@pytest.mark.gpu
def test_hip_batchnorm_moving_variance_closed_form():
    from aesrc2020_amd import _lib as L
    x, mv4, mv2, mm = _bn_case()
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    C = 4
    g, b = dev(np.ones(C)), dev(np.zeros(C))
    # 4-D path (resnet.py:25): sar_bn_stats_f32 with the Bessel-corrected moving variance
    mean, rstd, scale, shift = (torch.zeros(C, device="cuda") for _ in range(4))
    mmv, mvv = dev(np.zeros(C)), dev(np.ones(C))
    sums = torch.zeros(L.bn_workspace_doubles(C), dtype=torch.float64, device="cuda")
    L.call("sar_bn_stats_f32", dev(x), g, b, sums, mean, rstd, scale, shift, mmv, mvv, 2, C, 1e-3, 0.99, 1)
    np.testing.assert_allclose(mvv.cpu().numpy(), mv4, rtol=0, atol=1e-6)
    np.testing.assert_allclose(mmv.cpu().numpy(), mm, rtol=0, atol=1e-6)
    # 2-D path (model.py:286,289): Keras 2.2.x's n / (n - (1 + eps))
    y = torch.zeros(2, C, device="cuda"); mmv, mvv = dev(np.zeros(C)), dev(np.ones(C))
    L.call("sar_bn1d_fwd_f32", dev(x), g, b, y, mean, rstd, mmv, mvv, 2, C, 1e-3, 0.99)
    np.testing.assert_allclose(mvv.cpu().numpy(), mv2, rtol=0, atol=1e-6)
    np.testing.assert_allclose(mmv.cpu().numpy(), mm, rtol=0, atol=1e-6)
