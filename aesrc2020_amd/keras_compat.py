"""Keras checkpoint names -> this package's parameter names (the warm start of model.py:181-183, `raw_model='exp/xxx.h5'`).

The tensors themselves need no conversion: every parameter here has the Keras layout (conv HWIO, Dense (in, out), CuDNNGRU
kernel (in, 3H) / recurrent_kernel (H, 3H) / bias (6H) in z, r, h order).  Only the NAMES of an .h5 written by Keras 2.2.x
differ from the names of this package's .npz checkpoints, in two ways:
  * every weight name carries TensorFlow's ':0' output suffix                     'conv2d_7/kernel:0' -> 'conv2d_7/kernel'
  * Bidirectional(CuDNNGRU) puts each direction in an auto-numbered sub-layer     'CRNN/forward_cu_dnngru_3/kernel:0' -> 'CRNN/forward/kernel'
    (the number counts CuDNNGRU layers built in the session, so it is not stable between runs; the direction prefix is)
Nothing else is renamed: explicitly named layers (CNN_LIN, CRNN_LN, AR_BN1, gvlad_pool/centers, y_disc/W, ...) keep their names,
and the unnamed ResNet layers are auto-numbered conv2d_N / batch_normalization_N in construction order by both sides (a model
built first in a fresh Keras session, as train.py does).

h5py is not part of this image, so reading the file happens where h5py exists:
    python -m aesrc2020_amd.keras_compat exp/xxx.h5 exp/xxx.npz          (then raw_model='exp/xxx.npz')
`SARNetModel.load_weights` calls `load_h5` itself when h5py is importable and otherwise fails with that instruction.
"""
import re
import sys

import numpy as np

_BIDIR = re.compile(r"^(?P<layer>[^/]+)/(?P<dir>forward|backward)_[A-Za-z_]*gru(?:_\d+)?/(?P<w>kernel|recurrent_kernel|bias)$")


def keras_to_sar_name(name):
    """One Keras weight name (as stored in the 'weight_names' attribute of an .h5 layer group) -> the name used here."""
    if isinstance(name, bytes):
        name = name.decode("utf8")
    name = re.sub(r":\d+$", "", name)
    m = _BIDIR.match(name)
    if m:
        return "%s/%s/%s" % (m.group("layer"), m.group("dir"), m.group("w"))
    return name


def convert_keras_weights(weights):
    """{keras weight name: array} -> {name used here: float32 array}; raises on two Keras names mapping to one."""
    out = {}
    for k, v in weights.items():
        n = keras_to_sar_name(k)
        if n in out:
            raise ValueError("Keras weights %r and another one both map to %r" % (k, n))
        out[n] = np.asarray(v, dtype=np.float32)
    return out


def load_h5(path):
    """Read a Keras 2.2.x .h5 (model.save or save_weights) into {name used here: array}.  Needs h5py."""
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError("reading %s needs h5py, which this environment lacks.  Convert the checkpoint once where h5py is "
                           "installed:  python -m aesrc2020_amd.keras_compat %s %s   and pass the .npz as raw_model."
                           % (path, path, re.sub(r"\.(h5|hdf5|keras)$", "", str(path)) + ".npz")) from e
    weights = {}
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f              # model.save nests the weights, save_weights does not
        for layer in g.attrs["layer_names"]:
            lg = g[layer]
            for wn in lg.attrs["weight_names"]:
                weights[wn.decode("utf8") if isinstance(wn, bytes) else wn] = np.asarray(lg[wn])
    return convert_keras_weights(weights)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit("usage: python -m aesrc2020_amd.keras_compat model.h5 out.npz")
    d = load_h5(sys.argv[1])
    np.savez(sys.argv[2], **d)
    print("wrote %d tensors to %s" % (len(d), sys.argv[2]))
