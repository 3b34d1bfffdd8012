"""Host helpers mirroring the parts of /root/reference/utils.py the hot path's caller needs:
cal_descriptors (utils.py:156-159), label padding (utils.py:57-63), feature pad/trunc (utils.py:39-46), and the
synthetic batch used by bench.py / smoke in the exact dict format of utils.py:102-116."""
import math
import numpy as np

EOS_ID = 2


def cal_descriptors(T, D):
    def pool(x):
        return int(math.ceil(x / 2))
    for _ in range(5):
        T, D = pool(T), pool(D)
    return T * D


def feat_reshape(feat, max_len=1200):
    h, w = feat.shape
    if h >= max_len:
        return feat[:max_len]
    out = np.zeros((max_len, w), feat.dtype); out[:h] = feat
    return out


def text_ids_norm(ids, max_len):
    ids = list(ids)[:max_len]
    return ids + [EOS_ID] * (max_len - len(ids))


def synthetic_batch(input_shape, B, ctc_enable, disc_enable, accent_classes=8, bpe_classes=1000, max_ctc_len=72, seed=1234):
    """(inputs, targets) dicts: U[0,1) fbank with zero-padded tails (60-100 % of T kept), uniform accents,
    CTC labels in [0, V-2] with length U{4..16} clipped to what the encoder length admits."""
    rng = np.random.default_rng(seed)
    T, D, _ = input_shape
    x = rng.random((B, T, D, 1), dtype=np.float32)
    for b in range(B):
        x[b, int(rng.integers(int(math.ceil(0.6 * T)), T + 1)):] = 0
    onehot = np.eye(accent_classes, dtype=np.float32)[rng.integers(0, accent_classes, B)]
    inputs, targets = {"x_data": x}, {"y_accent": onehot}
    if disc_enable:
        inputs["x_accent"] = onehot; targets["y_disc"] = onehot
    if ctc_enable:
        L = cal_descriptors(T, D)
        lab = np.full((B, max_ctc_len), float(EOS_ID), np.float32); ll = np.zeros((B, 1), np.int32)
        for b in range(B):
            u = int(rng.integers(min(4, max(1, L // 3)), max(2, min(16, L // 2)) + 1)); u = min(u, max_ctc_len)
            lab[b, :u] = rng.integers(0, bpe_classes - 1, u); ll[b] = u
        inputs["x_ctc_label"] = lab; inputs["x_ctc_out_len"] = ll
        inputs["x_ctc_in_len"] = np.full((B, 1), L, np.int32); targets["y_ctc_loss"] = np.zeros(B, np.float32)
    return inputs, targets
