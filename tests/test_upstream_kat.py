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
