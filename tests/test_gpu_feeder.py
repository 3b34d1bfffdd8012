"""Data-feeder row: utils.data_loader / data_generator / Prefetcher with the MinMax+pad kernel
(csrc/feeder.hip) against (a) the golden vectors produced by the real sklearn MinMaxScaler (bit-exact: integer-style parity
for a float path, because the kernel follows sklearn's float32 operation order) and (b) the oracle's data_loader restatement."""
import os

import numpy as np
import pytest
import torch

from oracle import ops_np as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_minmax_pad_kernel_against_sklearn_golden():
    from aesrc2020_amd import utils as us
    g = np.load(os.path.join(GOLD, "feeder_golden.npz"))
    lens = g["lens"]; off = np.concatenate([[0], np.cumsum(lens)])
    feats = [g["frames"][off[i]:off[i + 1]] for i in range(len(lens))]
    out = us.feat_norm_pad(feats, int(g["max_len"]))
    torch.cuda.synchronize()
    assert out.shape == g["x_data"].shape and out.dtype == torch.float32 and out.is_cuda
    assert np.array_equal(out.cpu().numpy(), g["x_data"])                      # bit-exact vs real sklearn output


@pytest.mark.parametrize("B,D,max_len", [(1, 80, 17), (7, 80, 400), (130, 40, 50), (600, 13, 9), (3, 256, 33)])
def test_minmax_pad_kernel_shapes_and_edges(B, D, max_len):
    from aesrc2020_amd import utils as us
    rng = np.random.default_rng(B * 1000 + D)
    lens = rng.integers(1, 2 * max_len, B)
    feats = [(rng.standard_normal((n, D)) * 3 + 1).astype(np.float32) for n in lens]
    feats[0][:, 0] = 7.0
    got = us.feat_norm_pad(feats, max_len).cpu().numpy()
    want = np.float32(np.stack([O.feat_reshape(O.feat_norm(f), max_len) for f in feats]))[..., None]
    assert np.array_equal(got, want)
    # float64 pickles: sklearn works in f64 and the reference casts afterwards; the device path casts first
    f64 = [f.astype(np.float64) * 1.000001 for f in feats[:4]]
    got = us.feat_norm_pad(f64, max_len).cpu().numpy()
    want = np.float32(np.stack([O.feat_reshape(O.feat_norm(f), max_len) for f in f64]))[..., None]
    assert np.abs(got - want).max() < 2e-6
    with pytest.raises(ValueError):
        us.feat_norm_pad([feats[0], np.zeros((0, D), np.float32)], max_len)


def _dataset(tmp_path, n=12, pickled=True):
    from aesrc2020_amd import utils as us
    rng = np.random.default_rng(1)
    utts = ["utt%02d" % i for i in range(n)]
    feats = {u: (rng.random((int(rng.integers(60, 130)), 80)) * 9 - 2).astype(np.float32) for u in utts}
    data_dct = {}
    for u in utts:
        if pickled:
            p = str(tmp_path / (u + ".pkl")); us.save(p, feats[u]); data_dct[u] = p
        else:
            data_dct[u] = feats[u]
    acc = {u: str(int(rng.integers(0, 8))) for u in utts}
    trans = {u: [int(t) for t in rng.integers(3, 29, int(rng.integers(1, 4)))] for u in utts}
    return utts, feats, data_dct, acc, trans


def test_data_loader_matches_restatement(tmp_path):
    from aesrc2020_amd import utils as us
    utts, feats, data_dct, acc, trans = _dataset(tmp_path)
    kw = dict(ctc_enable=True, ar_enable=True, disc_enable=True, accent_dct=acc, trans_dct=trans, max_input_len=100,
              max_ctc_len=6, encoder_len=us.cal_descriptors(100, 80), bn=2)
    x, y = us.data_loader(utts, data_dct=data_dct, **kw)
    xr, yr = O.data_loader(utts, feats, **kw)
    assert set(x) == set(xr) and set(y) == set(yr)
    assert torch.is_tensor(x["x_data"]) and x["x_data"].is_cuda
    assert np.array_equal(x["x_data"].cpu().numpy(), xr["x_data"])
    for k in xr:
        if k != "x_data":
            assert np.array_equal(np.asarray(x[k]), xr[k]) and np.asarray(x[k]).dtype == xr[k].dtype, k
    for k in yr:
        assert np.array_equal(np.asarray(y[k]), yr[k]), k
    assert us.read_lines(_write_lines(tmp_path, utts)) == utts


def _write_lines(tmp_path, utts):
    p = str(tmp_path / "train.lst")
    with open(p, "wt") as f:
        f.write("\n".join(" %s " % u for u in utts) + "\n")
    return p


def test_fit_generator_with_device_feeder(tmp_path):
    """train.py:38-44 in miniature: data_generator -> fit_generator(max_queue_size=..) with the prefetch thread, validation
    on a data_loader batch (inference-mode evaluate), ReduceLROnPlateau / EarlyStopping wired like train.py:80-83."""
    import contextlib, io
    from aesrc2020_amd import utils as us
    from aesrc2020_amd.model import SAR_Net
    from aesrc2020_amd.callbacks import ReduceLROnPlateau, EarlyStopping
    utts, feats, data_dct, acc, trans = _dataset(tmp_path, n=13)
    T = 100
    kw = dict(ctc_enable=True, ar_enable=True, disc_enable=True, data_dct=data_dct, accent_dct=acc, trans_dct=trans,
              max_input_len=T, max_ctc_len=6, encoder_len=us.cal_descriptors(T, 80))
    with contextlib.redirect_stdout(io.StringIO()):
        model, tm = SAR_Net([T, 80, 1], ctc_enable=True, disc_enable=True, res_type="res18", res_filters=16, hidden_dim=32,
                            bpe_classes=30, max_ctc_len=6, mto="gvlad", vlad_clusters=4, metric_loss="cosface", lr=0.003)
    order0 = list(utts)
    gen = us.data_generator(utts, batch_size=4, **kw)
    dev_x, dev_y = us.data_loader(order0[:6], **kw)
    p0 = model.store.params.clone()
    with contextlib.redirect_stdout(io.StringIO()):
        hist = tm.fit_generator(generator=gen, steps_per_epoch=3, epochs=3, initial_epoch=0, validation_data=(dev_x, dev_y),
                                max_queue_size=4, workers=1, use_multiprocessing=False,       # Keras keywords without meaning here
                                callbacks=[ReduceLROnPlateau(monitor="val_y_accent_acc", factor=0.5, patience=1),
                                           EarlyStopping(monitor="val_y_accent_acc", patience=5)])
    assert hist.epoch == [0, 1, 2] and len(hist.history["loss"]) == 3 and "val_loss" in hist.history   # keras History
    with pytest.raises(TypeError):
        tm.fit_generator(generator=gen, steps_per_epoch=1, no_such_keyword=1)
    assert len(hist) == 3 and all(np.isfinite(h["loss"]) and np.isfinite(h["val_loss"]) for h in hist)
    assert {"val_y_accent_acc", "val_y_disc_acc", "y_ctc_loss_loss", "lr"} <= set(hist[-1])
    assert not torch.equal(p0, model.store.params)
    assert sorted(utts) == sorted(order0) and utts != order0                  # shuffled in place each epoch (utils.py:136)
