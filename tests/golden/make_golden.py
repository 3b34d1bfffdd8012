#!/usr/bin/env python
"""Generates the committed golden vectors under tests/golden/ from the CPU oracle (oracle/ops_np.py, oracle/model_torch.py).

The reference (pika-online/AESRC2020) ships no golden vectors and TensorFlow/Keras cannot be imported in the build
container, so these fixtures pin *our restatement* ("parity unpinned" w.r.t. the real Keras path): they freeze
the oracle's numbers so that (a) the oracle cannot drift silently (CPU test) and (b) the GPU parity tests have
data-only expectations that travel to the GPU box.      Usage: python tests/golden/make_golden.py

The upstream-HELD known answers (constants published in TensorFlow's ctc_loss_op_test.py, Keras' backend_test.py test_ctc and
the Keras categorical_crossentropy docstring -- numbers this script does not and must not produce) live next to these files in
tests/golden/upstream_kat.py, each with its source; tests/test_upstream_kat.py checks the oracle and the HIP kernels against
them.  They pin tf.nn.ctc_loss / K.ctc_batch_cost (loss to 1e-5, gradient to 2e-6) and the cross-entropy; everything else
here stays pinned only to the restatement.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ops_np as O          # noqa: E402
from oracle import model_torch as MT    # noqa: E402

MODEL_KW = dict(input_shape=[64, 80, 1], res_type="res18", res_filters=16, hidden_dim=32, mto="gvlad", vlad_clusters=4,
                ghost_clusters=2, disc_enable=True, ctc_enable=True, bpe_classes=20, max_ctc_len=6, metric_loss="circleloss")
# round 5: a second whole-graph fixture at the widths the SHIPPED kernels take (res_filters 32: csrc/conv_bf16.hip in split precision;
# hidden_dim 256: the MFMA recurrence) -- model_golden.npz's res_filters 16 runs the fp32-storage kernels in every mode
MODEL32_KW = dict(input_shape=[72, 80, 1], res_type="res18", res_filters=32, hidden_dim=256, mto="gvlad", vlad_clusters=6,
                  ghost_clusters=2, disc_enable=True, ctc_enable=True, bpe_classes=24, max_ctc_len=6, metric_loss="arcface")
MODEL32_GRADS = ("conv2d_1/kernel", "CTC_DS/bias", "conv2d_3/kernel", "conv2d_7/kernel", "conv2d_8/kernel", "conv2d_13/kernel", "batch_normalization_2/gamma", "batch_normalization_10/gamma", "batch_normalization_14/beta", "batch_normalization_9/beta",
                 "CNN_LIN/bias", "CNN_LIN_LN/gamma", "CRNN/forward/bias", "CRNN/backward/bias", "CTC_BIGRU/backward/bias", "ctc_pred/bias", "gvlad_center_assignment/kernel",
                 "gvlad_pool/centers", "AR_BN2/gamma", "y_accent/kernel", "y_disc/W")


def ops_vectors():
    rng = np.random.default_rng(2020)
    d = {}
    # K.ctc_batch_cost: repeated label, short input, infeasible item
    B, L, V, U = 4, 10, 7, 5
    p = O.softmax(rng.standard_normal((B, L, V)) * 2)
    lab = np.array([[1, 1, 3, 2, 2], [0, 5, 2, 2, 2], [4, 2, 2, 2, 2], [3, 3, 3, 3, 3]], np.float32)
    il = np.array([[10], [7], [10], [6]], np.int32); ll = np.array([[3], [2], [1], [5]], np.int32)
    loss, grad = O.ctc_batch_cost_fwd_bwd(p, lab, il, ll)
    d.update(ctc_probs=p, ctc_labels=lab, ctc_in_len=il, ctc_lab_len=ll, ctc_loss=loss, ctc_grad=grad)
    # VladPooling (gvlad)
    f = rng.standard_normal((3, 1, 9, 16)); s = rng.standard_normal((3, 1, 9, 6)) * 2; c = rng.standard_normal((6, 16))
    y, cache = O.vlad_fwd(f, s, c, 4, "gvlad"); dy = rng.standard_normal(y.shape)
    df, ds, dc = O.vlad_bwd(dy, cache)
    d.update(vlad_feat=f, vlad_score=s, vlad_centers=c, vlad_out=y, vlad_dout=dy, vlad_dfeat=df, vlad_dscore=ds, vlad_dcenters=dc)
    # margin heads
    x = rng.standard_normal((6, 16)); W = rng.standard_normal((16, 5)) * 0.3; yh = np.eye(5)[rng.integers(0, 5, 6)]; dp = rng.standard_normal((6, 5))
    d.update(head_x=x, head_W=W, head_y=yh, head_dp=dp)
    for mode, m in (("sphereface", 1.35), ("cosface", 0.3), ("arcface", 0.3)):
        pr, cache = O.margin_head_fwd(x, yh, W, mode, 30.0, m)
        dx, dW = O.margin_head_bwd(dp, cache)
        d.update({mode + "_probs": pr, mode + "_dx": dx, mode + "_dW": dW})
    sc = np.tanh(rng.standard_normal((6, 5)))
    cl, cds = O.circle_loss_fwd_bwd(yh, sc, 256.0, 0.3)
    pq = O.softmax(rng.standard_normal((6, 5)) * 5)
    ce, cdp = O.categorical_crossentropy_fwd_bwd(yh, pq)
    d.update(circle_s=sc, circle_loss=cl, circle_ds=cds, cce_p=pq, cce_loss=ce, cce_dp=cdp)
    return d


def feeder_vectors():
    """PINNED by third-party reference code: the expected outputs come from the real sklearn.preprocessing.MinMaxScaler
    (the class utils.py:5-6,35-36 calls; scikit-learn is importable in the build container), composed with the
    truncate / zero-pad rule of utils.py:39-46 -- not from the oracle."""
    from sklearn.preprocessing import MinMaxScaler
    import sklearn
    rng = np.random.default_rng(36)
    lens = [23, 40, 1, 57, 31]                      # shorter, equal, single-frame (all columns constant), longer than max_len
    max_len, D = 40, 80
    feats = [(rng.standard_normal((n, D)) * rng.uniform(0.5, 8) + rng.uniform(-20, 5)).astype(np.float32) for n in lens]
    feats[0][:, 3] = 1.25; feats[3][:, 79] = 0.0     # constant columns -> 0
    out = np.zeros((len(lens), max_len, D, 1), np.float32)
    for i, f in enumerate(feats):
        z = MinMaxScaler().fit_transform(f.copy())
        k = min(len(z), max_len)
        out[i, :k, :, 0] = z[:k]
    return dict(frames=np.concatenate(feats), lens=np.int32(lens), max_len=np.int32(max_len), x_data=out,
                sklearn_version=np.array(sklearn.__version__))


def model_vectors(kw=None, grads=None, B=3, batch_seed=11):
    cfg = MT.Cfg(**(kw or MODEL_KW))
    net = MT.TorchSARNet(cfg, seed=0)
    batch = MT.synthetic_batch(cfg, B, seed=batch_seed)
    out, total, parts, G = net.forward_backward(batch)
    # the weights are regenerated from the seed by oracle init (numpy PCG64 streams are version-stable); the fixture keeps a
    # per-tensor checksum so RNG/initialiser drift is detected instead of silently changing the expectations
    d = {"seed": np.array(0), "param_names": np.array(list(net.P)),
         "param_checksum": np.array([[float(v.sum()), float((v * v).sum())] for v in net.P.values()])}
    d.update({"batch/" + k: np.asarray(v) for k, v in batch.items()})
    d["out/y_accent"] = out["y_accent"].detach().numpy(); d["out/y_disc"] = out["y_disc"].detach().numpy()
    d["out/y_ctc_loss"] = out["y_ctc_loss"].detach().numpy(); d["out/ar"] = out["ar"].detach().numpy()
    d["loss/total"] = np.array(float(total))
    for k, v in parts.items():
        d["loss/" + k] = np.array(float(v))
    # gradients are stored for a spread of layers (first conv ... heads), minus the L2 term
    for k in grads or ("conv2d_1/kernel", "conv2d_5/kernel", "batch_normalization_3/gamma", "CNN_LIN/kernel", "CRNN/forward/recurrent_kernel",
                       "CTC_BIGRU/backward/kernel", "ctc_pred/kernel", "gvlad_center_assignment/kernel", "gvlad_pool/centers",
                       "AR_EMBEDDING/kernel", "y_accent/kernel", "y_disc/kernel"):
        g = G[k].numpy() - (2 * MT.L2 * net.P[k].numpy() if k in net.reg else 0.0)
        d["grad/" + k] = g
    return d


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "ops_golden.npz"), **ops_vectors())
    np.savez_compressed(os.path.join(HERE, "model_golden.npz"), **model_vectors())
    np.savez_compressed(os.path.join(HERE, "feeder_golden.npz"), **feeder_vectors())
    np.savez_compressed(os.path.join(HERE, "model32_golden.npz"), **model_vectors(MODEL32_KW, MODEL32_GRADS, B=4, batch_seed=12))
    for f in ("ops_golden.npz", "model_golden.npz", "model32_golden.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
