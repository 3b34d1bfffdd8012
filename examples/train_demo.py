#!/usr/bin/env python
"""The reference's train.py (train.py:9-118) on the MI355X path, end to end, on synthetic data: pickled fbank utterances on
disk -> us.data_generator (MinMax + padding on the GPU) -> mdl.SAR_Net -> fit_generator with the reference's callbacks ->
per-epoch model.save -> warm start of a second model from the checkpoint (raw_model=, model.py:181-183) -> predict.

    python examples/train_demo.py [workdir]          (needs one MI355X; ~20 s)

Only the two import lines differ from the reference script."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aesrc2020_amd import utils as us                       # reference: import utils as us
from aesrc2020_amd import model as mdl                      # reference: import model as mdl
from aesrc2020_amd.callbacks import ReduceLROnPlateau, CSVLogger, EarlyStopping, Callback   # reference: keras.callbacks

# ---- hyper-parameters as in train.py:49-78 (sizes reduced for a demo)
MAX_INPUT_LEN, FEAT_DIM, MAX_CTC_LEN = 200, 80, 16
RES_TYPE, RES_FILTERS, HIDDEN_DIM = "res18", 32, 128
BPE_CLASSES, ACCENT_CLASSES = 100, 8
BATCH_SIZE, EPOCHS, N_UTT = 16, 3, 96


def make_dataset(root, rng):
    """scp-like dicts of train.py:88-104: utt -> feature pickle, utt -> accent id, utt -> BPE ids."""
    data, accent, trans = {}, {}, {}
    for i in range(N_UTT):
        utt = "utt%04d" % i
        feat = rng.standard_normal((int(rng.integers(120, 260)), FEAT_DIM)).astype(np.float32) * 3 - 1
        path = os.path.join(root, utt + ".pkl"); us.save(path, feat)
        data[utt] = path; accent[utt] = str(int(rng.integers(0, ACCENT_CLASSES)))
        trans[utt] = [int(t) for t in rng.integers(3, BPE_CLASSES - 1, int(rng.integers(2, 7)))]
    return data, accent, trans


class SaveEveryEpoch(Callback):                              # train.py:31-35
    def __init__(self, model, out_dir):
        self.m, self.out = model, out_dir

    def on_epoch_end(self, epoch, logs=None):
        self.m.save("%s/%03d.h5" % (self.out, epoch))


def main():
    work = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="sarnet_demo_")
    os.makedirs(work, exist_ok=True)
    rng = np.random.default_rng(0)
    data, accent, trans = make_dataset(work, rng)
    utts = sorted(data)
    train_lst, dev_lst = utts[:80], utts[80:]
    enc_len = us.cal_descriptors(MAX_INPUT_LEN, FEAT_DIM)
    kw = dict(ctc_enable=True, ar_enable=True, disc_enable=True, data_dct=data, accent_dct=accent, trans_dct=trans,
              max_input_len=MAX_INPUT_LEN, max_ctc_len=MAX_CTC_LEN, encoder_len=enc_len, accent_classes=ACCENT_CLASSES)
    model, train_model = mdl.SAR_Net(input_shape=[MAX_INPUT_LEN, FEAT_DIM, 1], ctc_enable=True, ar_enable=True, disc_enable=True,
                                     res_type=RES_TYPE, res_filters=RES_FILTERS, hidden_dim=HIDDEN_DIM, bn_dim=0,
                                     bpe_classes=BPE_CLASSES, accent_classes=ACCENT_CLASSES, max_ctc_len=MAX_CTC_LEN,
                                     mto="gvlad", vlad_clusters=8, ghost_clusters=2, metric_loss="circleloss", margin=0.3,
                                     raw_model=None, lr=0.003, gpus=1)
    dev_x, dev_y = us.data_loader(dev_lst, **kw)
    callbacks = [ReduceLROnPlateau(monitor="val_y_accent_acc", factor=0.3, patience=2),
                 EarlyStopping(monitor="val_y_accent_acc", patience=10),
                 CSVLogger(os.path.join(work, "train.csv")), SaveEveryEpoch(model, work)]
    train_model.fit_generator(generator=us.data_generator(train_lst, batch_size=BATCH_SIZE, **kw),
                              steps_per_epoch=len(train_lst) // BATCH_SIZE, epochs=EPOCHS, callbacks=callbacks, initial_epoch=0,
                              validation_data=(dev_x, dev_y), max_queue_size=20)
    # warm start (the ASR-init -> AR fine-tune recipe of model.py:181-183), inference graph with a free time axis
    infer, _ = mdl.SAR_Net(input_shape=[MAX_INPUT_LEN, FEAT_DIM, 1], ctc_enable=True, ar_enable=True, disc_enable=True,
                           res_type=RES_TYPE, res_filters=RES_FILTERS, hidden_dim=HIDDEN_DIM, bpe_classes=BPE_CLASSES,
                           accent_classes=ACCENT_CLASSES, max_ctc_len=MAX_CTC_LEN, mto="gvlad", vlad_clusters=8, ghost_clusters=2,
                           metric_loss="circleloss", margin=0.3, raw_model="%s/%03d.h5" % (work, EPOCHS - 1), mode="predict")
    y_accent, y_disc, _ = infer.predict({k: v for k, v in dev_x.items()}, batch_size=8)
    ref_accent, _, _ = model.predict({k: v for k, v in dev_x.items()}, batch_size=8)
    assert np.allclose(y_accent, ref_accent, atol=1e-5), "checkpoint round trip changed the predictions"
    print("demo OK: %d dev utterances, accent accuracy %.2f, checkpoints in %s" %
          (len(dev_lst), float((y_accent.argmax(1) == dev_y["y_accent"].argmax(1)).mean()), work))


if __name__ == "__main__":
    main()
