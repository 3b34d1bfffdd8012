"""Host side of the data-feeding boundary, mirroring /root/reference/utils.py: file helpers (utils.py:14-27), feature
normalisation + padding (35-46), label padding (57-63), `data_loader` / `data_generator` (71-154) with the reference's
signatures and batch-dict format (102-116), `cal_descriptors` (156-159), plus the synthetic batch used by bench.py / smoke
(SURVEY 8d).

What is different from the reference: the per-utterance MinMax scaling + zero padding -- the only arithmetic on this row --
runs on the MI355X (csrc/feeder.hip, `sar_feat_minmax_pad_f32`) on the ragged batch uploaded through pinned memory, so
`x_data` comes back as a device tensor that `train_on_batch` / `fit_generator` consume without another copy.  There is no
host implementation of that arithmetic here: without the HIP library these functions raise."""
import functools
import math
import pickle
import queue
import threading
from random import shuffle

import numpy as np

EOS_ID = 2


def cal_descriptors(T, D):
    def pool(x):
        return int(math.ceil(x / 2))
    for _ in range(5):
        T, D = pool(T), pool(D)
    return T * D


# ---- FILE OPT (utils.py:14-27)
def save(path, arr):
    with open(path, "wb") as f:
        pickle.dump(arr, f)


def load(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def read_lines(path):
    with open(path, "rt") as f:
        return [line.strip() for line in f.readlines()]


def to_categorical(y, num_classes):
    """keras.utils.np_utils.to_categorical for one integer label (utils.py:92)."""
    out = np.zeros(num_classes, np.float32); out[int(y)] = 1.0
    return out


# ---- FEATURE OPT (utils.py:35-46), device side
def feat_norm_pad(feats, max_len=1200, stream=None):
    """feat_reshape(feat_norm(f), max_len) for a list of (n_i, D) utterances -> device tensor (B, max_len, D, 1) f32.
    The ragged batch goes through one pinned staging buffer and one H2D copy; MinMax scaling (per utterance, per feature
    column, over all its frames) and truncation / zero padding run in `sar_feat_minmax_pad_f32`."""
    import torch
    from . import _lib
    B = len(feats)
    lens = [int(np.shape(f)[0]) for f in feats]
    if B == 0 or min(lens) == 0:
        # sklearn's MinMaxScaler raises on an empty utterance (utils.py:36)
        raise ValueError("Found array with 0 sample(s) while a minimum of 1 is required by MinMaxScaler.")
    D = int(np.shape(feats[0])[1])
    off = np.zeros(B + 1, np.int32); off[1:] = np.cumsum(lens)
    host = torch.empty((int(off[-1]), D), dtype=torch.float32).pin_memory()
    hv = host.numpy()
    for i, f in enumerate(feats):
        hv[off[i]:off[i + 1]] = f                                           # casts float64 pickles to f32
    st = stream or torch.cuda.current_stream()
    with torch.cuda.stream(st):
        frames = host.cuda(non_blocking=True)
        offs = torch.from_numpy(off).pin_memory().cuda(non_blocking=True)
        out = torch.empty((B, max_len, D, 1), dtype=torch.float32, device="cuda")
        _lib.call("sar_feat_minmax_pad_f32", frames, offs, out, B, max_len, D, stream=st.cuda_stream)
    return out


def feat_reshape(feat, max_len=1200):
    """utils.py:39-46 (pure data movement; kept for callers that pad already-normalised features)."""
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


# ---- Generator: the batch contract of utils.py:71-154 (keys, dtypes, shapes), built batch-wise
def _ctc_targets(transcripts, width, encoder_len):
    """Token-id lists -> the three CTC inputs of a batch: labels (B, width) float32 padded with EOS_ID (K.ctc_batch_cost takes
    float labels, utils.py:107), label lengths clipped to `width` and encoder lengths, both (B, 1) int32."""
    n = len(transcripts)
    lens = np.fromiter((min(len(t), width) for t in transcripts), dtype=np.int32, count=n)
    labels = np.full((n, width), EOS_ID, dtype=np.float32)
    keep = np.arange(width)[None, :] < lens[:, None]
    if lens.sum():
        labels[keep] = np.concatenate([np.asarray(t[:k], dtype=np.float32) for t, k in zip(transcripts, lens) if k])
    return labels, lens[:, None], np.full((n, 1), encoder_len, dtype=np.int32)


def data_loader(lst, ctc_enable=False, ar_enable=False, disc_enable=False, data_dct=None, accent_dct=None, trans_dct=None,
                max_input_len=1200, max_ctc_len=72, encoder_len=100, accent_classes=8, bn=0, stream=None):
    """One batch for the utterance ids in `lst`, as (inputs, targets) dicts with the reference's keys (utils.py:102-116):
    x_data (B, max_input_len, D, 1) float32 -- here a device tensor straight out of feat_norm_pad (MinMax + pad on the GPU) --,
    x_ctc_in_len / x_ctc_out_len (B, 1) int32, x_ctc_label (B, max_ctc_len) float32, y_ctc_loss zeros(B), y_accent / x_accent /
    y_disc [/ y_disc_bn when `bn`] one-hot (B, accent_classes) float32.  `data_dct[utt]` is a pickle path or an array."""
    feats = [load(f) if isinstance(f, str) else f for f in (data_dct[utt] for utt in lst)]
    inputs, targets = {"x_data": feat_norm_pad(feats, max_input_len, stream=stream)}, {}
    if ctc_enable:
        if trans_dct:
            labels, out_len, in_len = _ctc_targets([trans_dct[utt] for utt in lst], max_ctc_len, encoder_len)
        else:                                     # the reference appends nothing without transcripts: empty arrays, one trailing axis
            labels, out_len, in_len = np.zeros((0,), np.float32), np.zeros((0, 1), np.int32), np.zeros((0, 1), np.int32)
        inputs.update(x_ctc_in_len=in_len, x_ctc_out_len=out_len, x_ctc_label=labels)
        targets["y_ctc_loss"] = np.zeros(len(lst))
    # one-hot accent rows exist only when the AR task is on and labels were given (utils.py:99-100); the discriminator keys
    # reuse them whatever they are
    if ar_enable and accent_dct:
        onehot = np.eye(accent_classes, dtype=np.float32)[np.fromiter((int(accent_dct[utt]) for utt in lst), dtype=np.int64, count=len(lst))]
    else:
        onehot = np.zeros((0,), np.float32)
    if ar_enable:
        targets["y_accent"] = onehot
    if disc_enable:
        inputs["x_accent"] = onehot
        targets["y_disc"] = onehot
        if bn:
            targets["y_disc_bn"] = onehot
    return inputs, targets


def data_generator(lst, ctc_enable=False, ar_enable=False, disc_enable=False, batch_size=32, data_dct=None, accent_dct=None,
                   trans_dct=None, max_input_len=1200, max_ctc_len=72, encoder_len=100, accent_classes=8, bn=0, stream=None):
    """Endless stream of batches (utils.py:120-154): each epoch reshuffles the caller's list IN PLACE and covers its first
    len(lst)//batch_size full batches; the remainder of an epoch is dropped."""
    make = functools.partial(data_loader, ctc_enable=ctc_enable, ar_enable=ar_enable, disc_enable=disc_enable, data_dct=data_dct,
                             accent_dct=accent_dct, trans_dct=trans_dct, max_input_len=max_input_len, max_ctc_len=max_ctc_len,
                             encoder_len=encoder_len, accent_classes=accent_classes, bn=bn, stream=stream)
    while True:
        shuffle(lst)
        for start in range(0, len(lst) - batch_size + 1, batch_size):
            yield make(lst[start:start + batch_size])


class Prefetcher:
    """What Keras' GeneratorEnqueuer does for fit_generator(max_queue_size=20, workers=1) (train.py:38-44): one background
    thread runs the generator ahead of the training loop.  The thread issues its H2D copy + feeder kernel on its own HIP
    stream and hands over (batch, event); the consumer makes the training stream wait on the event -- the upload of batch
    i+1 overlaps the (graph-replayed) step of batch i."""

    def __init__(self, generator, max_queue_size=10):
        import torch
        self.q = queue.Queue(maxsize=max(1, int(max_queue_size)))
        self.gen = generator
        self.stop = threading.Event()
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _run(self):
        import torch
        try:
            ctx = torch.cuda.stream(self.stream) if self.stream is not None else None
            if ctx:
                ctx.__enter__()
            while not self.stop.is_set():
                item = next(self.gen)
                ev = None
                if self.stream is not None:
                    ev = torch.cuda.Event(); ev.record(self.stream)
                while not self.stop.is_set():
                    try:
                        self.q.put((item, ev), timeout=0.1); break
                    except queue.Full:
                        continue
        except StopIteration:
            self.q.put((None, None))
        except BaseException as e:                                        # surface feeder errors in the training thread
            self.q.put((e, None))

    def __iter__(self):
        return self

    def __next__(self):
        import torch
        item, ev = self.q.get()
        if isinstance(item, BaseException):
            raise item
        if item is None:
            raise StopIteration
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        return item

    def close(self):
        self.stop.set()
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        self.th.join(timeout=5)
