"""SAR_Net: the multi-task speech-accent-recognition graph of /root/reference/model.py:204-371, MI355X-native.

`SAR_Net(**kw) -> (model, train_model)` keeps the reference signature, layer names, weight layouts, batch-dict
format (utils.py:102-116), loss assembly (model.py:344-369 incl. the weight overwrite at line 361) and the Keras
model methods train.py uses (fit_generator / train_on_batch / save / load_weights / summary / get_layer / predict).
The step itself is a static plan of hand-written HIP kernel launches (forward, explicit backward, fused L2+Adam)
captured into one HIP graph; data parallelism is one process per GPU with an RCCL all-reduce of the flat gradient
buffer (parallel.py) instead of keras.utils.multi_gpu_model (model.py:193-194)."""
import os

import numpy as np
import torch

from . import _lib
from . import losses as ls
from . import VLAD as vd
from .engine import ParamStore, store_scope, orthogonal, glorot_uniform
from .layers import Dense, LayerNorm, BiGRU, dense_ln_forward, BatchNorm1d, GlobalAveragePooling1D, L2Normalize, Layer
from .resnet import resnet18_, resnet34_, resnet50_, resnet101_, resnet152_
from .parallel import DataParallel

_call = _lib.call


class Input:
    """Shape holder standing in for keras Input (model.py:230-235)."""

    def __init__(self, shape, name=None, dtype="float32"):
        self.shape, self.name, self.dtype = tuple(shape), name, dtype


# ---- layer helpers with the reference's names (model.py:29-53) -------------------------------------------------
def BN(dim, name=None):
    return BatchNorm1d(dim, name=name)


def LN(dim, name=None):
    return LayerNorm(dim, name=name)


def DS(in_dim, hidden, activation, rgr=1e-4, use_bias=True, name=None):
    return Dense(in_dim, hidden, activation=activation, use_bias=use_bias, name=name, l2=rgr, init="he_normal")


def BIGRU(in_dim, hidden, seq=True, rgr=1e-4, name=None):
    return BiGRU(in_dim, hidden, seq=seq, name=name, l2=rgr)


class _AssignConv(Dense):
    """1x1 Conv2D cluster assignment (model.py:89-95 / 99-105): orthogonal init, l2(1e-4) on kernel+bias.
    Stored with the Keras kernel shape (1,1,H,K+G); computed as a GEMM over the (B*L, H) rows."""

    def __init__(self, in_dim, units, name):
        Layer.__init__(self, name)
        rng = self.store.rng
        w = orthogonal(rng, (in_dim, units)) if in_dim >= units else orthogonal(rng, (units, in_dim)).T
        self.kernel = self.store.add(name + "/kernel", (1, 1, in_dim, units), w, 1e-4)
        self.bias = self.store.add(name + "/bias", (units,), np.zeros(units), 1e-4)
        self.in_dim, self.units, self.activation = in_dim, units, None
        self.output_shape = (None, units)


class _VladBlock:
    """vlad() of model.py:82-109: assignment conv + VladPooling."""

    def __init__(self, hidden, aggregation, vlad_clusters, ghost_clusters):
        g = ghost_clusters if aggregation == "gvlad" else 0
        self.assign = _AssignConv(hidden, vlad_clusters + g, aggregation + "_center_assignment")
        self.pool = vd.VladPooling(k_centers=vlad_clusters, g_centers=g, mode=aggregation, name=aggregation + "_pool", feat_dim=hidden)
        self.out_dim = vlad_clusters * hidden

    def forward(self, x, B, L):
        score = self.assign.forward(x)
        return self.pool.forward(x.view(B, L, -1), score.view(B, L, -1))

    def backward(self, dout):
        dfeat, dscore = self.pool.backward(dout)
        self.assign.backward(dscore, dx_out=dfeat.view(-1, dfeat.shape[-1]), accumulate_dx=True)
        return dfeat.view(-1, dfeat.shape[-1])


class SARNetModel:
    """The object SAR_Net returns (both as `model` and `train_model`)."""

    def __init__(self, cfg):
        self.cfg = cfg
        c = cfg
        self.store = ParamStore(seed=c["seed"])
        self.name = c["name"] or "model"
        T, D, _ = c["input_shape"]
        H, A = c["hidden_dim"], c["accent_classes"]
        if H < 4 or H % 4:
            raise ValueError("hidden_dim must be a positive multiple of 4 (16-byte rows of the Dense / GRU kernels; got %r; the reference's "
                             "default is 256, model.py:204).  Up to 256 the recurrence runs on the persistent kernels (256: on MFMA), "
                             "above that one GEMM + one gate launch per time step" % (H,))
        with store_scope(self.store):
            inputs = Input(shape=c["input_shape"], name="x_data")
            fn = {"res18": resnet18_, "res34": resnet34_, "res50": resnet50_, "res101": resnet101_, "res152": resnet152_}
            if c["res_type"] not in fn:
                # the reference only prints here (model.py:250-251) and then fails on the undefined `cnn`
                print("======= ERROR: please specify cnn in res-[18,34,50,101,152] ======")
                raise ValueError("res_type")
            self.trunk = fn[c["res_type"]](inputs, filters=c["res_filters"])
            if getattr(self.trunk, "half", False) or getattr(self.trunk, "x3", False) or _lib.get_gemm_math() == "bf16x3":
                # fp16 operands can overflow (csrc/optim.hip): the fp16 / split-precision trunk's activation gradients, and -- whatever the trunk
                # runs on (res_filters off {32, 64}, SAR_TRUNK_DTYPE=fp32 / bf16) and whatever hidden_dim is -- the split-precision GEMM mode's
                # backward: Dense dX, the Bi-GRU pair dX and the grouped weight gradients are single-pass fp16 under 64 x batch
                # (ParamStore.bwd_gemm_scale), and at hidden_dim 256 the recurrence is the fp16 MFMA form (layers.BIGRU: msfx "f16").  Without
                # the guard an overflow would reach Adam's moments for good
                self.store.set_adam_guard(True)
            Tp, Dp = self.trunk.out_hw
            self.L = Tp * Dp
            C = self.trunk.out_c
            self.CNN_LIN = DS(C, H, "tanh", name="CNN_LIN"); self.CNN_LIN_LN = LN(H, name="CNN_LIN_LN")
            self.CRNN = BIGRU(H, H, name="CRNN"); self.CRNN_LN = LN(2 * H, name="CRNN_LN")
            if c["ctc_enable"]:
                self.CTC_BIGRU = BIGRU(2 * H, H, name="CTC_BIGRU"); self.CTC_BIGRU_LN = LN(2 * H, name="CTC_BIGRU_LN")
                self.CTC_DS = DS(2 * H, H, "tanh", name="CTC_DS"); self.CTC_DS_LN = LN(H, name="CTC_DS_LN")
                self.ctc_pred = DS(H, c["bpe_classes"], "softmax", name="ctc_pred")
            self._fused_accent = False
            if c["ar_enable"]:
                self.AR_DS = DS(2 * H, H, "tanh", name="AR_DS"); self.AR_DS_LN = LN(H, name="AR_DS_LN")
                mto = c["mto"]
                if mto == "avg":
                    self.merge = GlobalAveragePooling1D(H, name="AR_MERGE"); pooled = H
                elif mto == "bigru":
                    self.merge = BIGRU(H, H, seq=False, name="AR_MERGE"); pooled = 2 * H
                elif mto in ("vlad", "gvlad"):
                    self.merge = _VladBlock(H, mto, c["vlad_clusters"], c["ghost_clusters"]); pooled = self.merge.out_dim
                else:
                    print("Please specify avg/bigru/vlad/gvlad ..")           # model.py:137-138 (exit(1) there)
                    raise SystemExit(1)
                self.AR_BN1 = BN(pooled, name="AR_BN1")
                self.AR_EMBEDDING = DS(pooled, H, None, name="AR_EMBEDDING")
                self.AR_BN2 = BN(H, name="AR_BN2")
                self.AR_CF_DS1 = DS(H, 64, "relu", name="AR_CF_DS1"); self.AR_CF_DS2 = DS(64, 64, "relu", name="AR_CF_DS2")
                self.y_accent = DS(64, A, "softmax", name="y_accent")
                # one launch for the three layers + cross-entropy + their backward where the sizes fit its LDS plan
                self._fused_accent = A <= 32 and H <= 256 and os.environ.get("SAR_FUSED_ACCENT", "1") == "1"
                self.disc = None
                if c["disc_enable"]:
                    ml = c["metric_loss"]
                    if ml == "softmax":
                        self.disc = DS(H, A, "softmax", use_bias=False, name="y_disc")
                    elif ml in ("sphereface", "cosface", "arcface"):
                        cls = {"sphereface": ls.SphereFace, "cosface": ls.CosFace, "arcface": ls.ArcFace}[ml]
                        self.disc = cls(n_classes=A, m=c["margin"], name="y_disc", in_dim=H)
                    elif ml == "circleloss":
                        self.disc_norm = L2Normalize(H, name="y_disc_l2norm")
                        self.disc = Dense(H, A, activation=None, use_bias=False, name="y_disc", l2=0.0, init="glorot_uniform",
                                          constraint="unit_norm")
                    else:
                        raise ValueError("unknown metric_loss %r (the reference's disc_loss returns None, model.py:164-165)" % ml)
                if c["bn_dim"] and not c["disc_enable"]:
                    # model.py:337-338 appends the undefined `bn` to the outputs (NameError in the reference)
                    raise ValueError("bn_dim requires disc_enable=True (model.py:312, 337-338)")
                if c["disc_enable"] and c["bn_dim"]:                          # visual bottleneck head, model.py:312-322
                    nb = c["bn_dim"]
                    self.AR_BN_DS = DS(H, 64, "relu", name="AR_BN_DS"); self.AR_BN3 = BN(64, name="AR_BN3")
                    self.bottleneck = DS(64, nb, None, name="bottleneck"); self.AR_BN4 = BN(nb, name="AR_BN4")
                    if ml == "softmax":
                        self.disc_bn = DS(nb, A, "softmax", use_bias=False, name="y_disc_bn")
                    elif ml == "circleloss":
                        self.disc_bn_norm = L2Normalize(nb, name="y_disc_bn_l2norm")
                        self.disc_bn = Dense(nb, A, activation=None, use_bias=False, name="y_disc_bn", l2=0.0,
                                             init="glorot_uniform", constraint="unit_norm")
                    else:
                        self.disc_bn = cls(n_classes=A, m=c["margin"], name="y_disc_bn", in_dim=nb)
        self.store.finalize()
        self.dp = DataParallel(self.store, trunk_splits=self.trunk.bucket_split_names())
        # SAR_Net(gpus=N) (model.py:193-194, multi_gpu_model): the caller keeps feeding GLOBAL batches, every rank takes its
        # contiguous slice (same slicing as multi_gpu_model's Lambda layers)
        self.shard_global = int(c.get("gpus") or 1) > 1 and self.dp.world > 1
        self.store.set_adam(c["lr"], grad_scale=1.0 / self.dp.world)
        self.bufs = self.store.bufs
        self._loss_weights()
        self.cce_accent = ls.CategoricalCrossentropy(self.bufs, "loss_y_accent")
        self.cce_disc = ls.CategoricalCrossentropy(self.bufs, "loss_y_disc")
        self.circle = ls.CircleLoss(self.bufs, "loss_circle")
        self.cce_disc_bn = ls.CategoricalCrossentropy(self.bufs, "loss_y_disc_bn")
        self.graph = None; self.graph_B = None; self.use_graph = c["use_graph"]; self._eager_done = 0
        self.stop_training = False
        print(self.loss_weights)                                               # model.py:370

    # ------------------------------------------------------------------ loss assembly (model.py:344-369)
    def _loss_weights(self):
        c = self.cfg
        alpha, beta = 0.4, 0.01
        lw = {}
        if c["ar_enable"]:
            lw["y_accent"] = beta if c["disc_enable"] else 1.0
            if c["disc_enable"]:
                lw["y_disc"] = 1 - alpha if c["ctc_enable"] else 1.0
        if c["ctc_enable"]:
            lw["y_ctc_loss"] = alpha if c["disc_enable"] else 1.0
            lw["y_ctc_loss"] = 1 - alpha if not c["disc_enable"] else beta      # the reference's overwrite (line 361)
        if c["bn_dim"]:
            lw["y_disc_bn"] = 0.1                                               # model.py:363-366
        self.loss_weights = lw
        self.output_names = [n for n in ("y_accent", "y_disc", "y_ctc_loss", "y_disc_bn") if n in lw]   # model.py:327-338
        self.metrics_names = ["loss"] + [n + "_loss" for n in self.output_names] + \
                             [n + "_acc" for n in self.output_names if n != "y_ctc_loss"]

    # ------------------------------------------------------------------ inputs
    def _bind_inputs(self, B, T=None):
        c = self.cfg
        self.store._zero_list = {}; self.store._zero_table = None; self.store._zero_rows = set()   # buffers of the old batch shape are about to be replaced
        self.store._zeroed_this_step = False; self._copy_tables = {}; self._copy_tables_for = None
        D = c["input_shape"][1]
        T = int(T if T is not None else c["input_shape"][0])
        if T != self.trunk.in_hw[0]:
            if c["mode"] == "train":
                raise ValueError("x_data has %d frames but the train-mode graph was built for %d (model.py:228-229); "
                                 "build with mode != 'train' for a variable-length time axis" % (T, self.trunk.in_hw[0]))
            Tp, Dp = self.trunk.plan(T, D)                                    # model.py:230-231: Input(shape=[None, D, 1])
            self.L = Tp * Dp
        b = self.bufs
        self.B = B; self.T = T
        self.store.batch = B                                         # (engine.bwd_gemm_scale: the gradient scale of the fp16 backward products)
        self.in_x = b.get("in/x_data", (B, T, D, 1))
        A = c["accent_classes"]
        self.in_y_accent = b.get("in/y_accent", (B, A))
        if c["disc_enable"]:
            self.in_x_accent = b.get("in/x_accent", (B, A)); self.in_y_disc = b.get("in/y_disc", (B, A))
            if c["bn_dim"]:
                self.in_y_disc_bn = b.get("in/y_disc_bn", (B, A))
        if c["ctc_enable"]:
            self.in_ctc_label = b.get("in/x_ctc_label", (B, c["max_ctc_len"]))
            self.in_ctc_in_len = b.get("in/x_ctc_in_len", (B,), torch.int32)
            self.in_ctc_out_len = b.get("in/x_ctc_out_len", (B,), torch.int32)
            self.ctc_ws_bytes = _lib.ctc_workspace_bytes(B, self.L, c["max_ctc_len"])
            self.ctc_ws = b.get("ctc/ws", (self.ctc_ws_bytes // 4,))
            self.ctc_loss = b.get("ctc/loss", (B,)); self.ctc_dprobs = b.get("ctc/dprobs", (B * self.L, c["bpe_classes"]))
        # constant seed gradients: d(total)/d(per-sample loss) = weight / B (Keras batch-mean)
        self.seed = {}
        for k, w in self.loss_weights.items():
            t = b.get("seed/" + k, (B,)); t.fill_(w / B); self.seed[k] = t

    def _upload(self, inputs, targets):
        c = self.cfg
        if self.shard_global and self.store.training:
            from .parallel import shard_batch
            inputs = shard_batch(inputs, self.dp.rank, self.dp.world)
            targets = shard_batch(targets, self.dp.rank, self.dp.world) if targets else targets
        x = inputs["x_data"]
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(np.asarray(x, np.float32)))
        B, T = int(x.shape[0]), int(x.shape[1])
        if getattr(self, "B", None) != B or getattr(self, "T", None) != T:
            self._bind_inputs(B, T)
            self.graph = None; self._eager_done = 0
        self.in_x.copy_(x.reshape(self.in_x.shape))                           # H2D, or D2D for the device feeder's x_data
        if x.is_cuda:
            x.record_stream(torch.cuda.current_stream())                      # produced on the feeder's stream (utils.Prefetcher)
        tg = targets or {}
        A = c["accent_classes"]
        zeros_a = np.zeros((B, A), np.float32)                                # predict(): labels may be absent
        if c["ar_enable"]:
            ya = tg.get("y_accent", inputs.get("x_accent", zeros_a))
            self.in_y_accent.copy_(torch.from_numpy(np.asarray(ya, np.float32)))
        if c["disc_enable"]:
            xa = inputs.get("x_accent", zeros_a)
            self.in_x_accent.copy_(torch.from_numpy(np.asarray(xa, np.float32)))
            self.in_y_disc.copy_(torch.from_numpy(np.asarray(tg.get("y_disc", xa), np.float32)))
            if c["bn_dim"]:
                self.in_y_disc_bn.copy_(torch.from_numpy(np.asarray(tg.get("y_disc_bn", xa), np.float32)))
        if c["ctc_enable"]:
            if "x_ctc_label" in inputs:
                lab = np.asarray(inputs["x_ctc_label"], np.float32)
                il = np.asarray(inputs["x_ctc_in_len"]).reshape(-1).astype(np.int32)
                ol = np.asarray(inputs["x_ctc_out_len"]).reshape(-1).astype(np.int32)
            else:                                                             # predict(): dummy 1-symbol targets
                lab = np.zeros((B, c["max_ctc_len"]), np.float32)
                il = np.full(B, self.L, np.int32); ol = np.ones(B, np.int32)
            self._check_ctc_feasible(lab, il, ol)
            self.in_ctc_label.copy_(torch.from_numpy(np.ascontiguousarray(lab)))
            self.in_ctc_in_len.copy_(torch.from_numpy(il)); self.in_ctc_out_len.copy_(torch.from_numpy(ol))

    @staticmethod
    def _check_ctc_feasible(lab, il, ol):
        """tf.nn.ctc_loss raises 'Not enough time for target transition sequence' (SURVEY App. A); so do we."""
        for b in range(lab.shape[0]):
            u = int(min(ol[b], lab.shape[1])); l = lab[b, :u].astype(np.int64)
            need = u + int((l[1:] == l[:-1]).sum())
            if need > il[b]:
                raise ValueError("Not enough time for target transition sequence (required: %d, available: %d) in batch item %d"
                                 % (need, il[b], b))

    def set_device_batch(self, dev):
        """Device-resident batch of THIS rank (dict of torch tensors, same keys as the host batch dict): D2D copies only; never
        sharded, whatever `gpus` says (bench.py keeps 64 utterances per GPU)."""
        B, T = dev["x_data"].shape[0], dev["x_data"].shape[1]
        if getattr(self, "B", None) != B or getattr(self, "T", None) != T:
            self._bind_inputs(B, T); self.graph = None; self._eager_done = 0
        c = self.cfg
        pairs = [(self.in_x, dev["x_data"])]
        if c["ar_enable"]:
            pairs.append((self.in_y_accent, dev["y_accent"]))
        if c["disc_enable"]:
            pairs += [(self.in_x_accent, dev["x_accent"]), (self.in_y_disc, dev["y_disc"])]
            if c["bn_dim"]:
                pairs.append((self.in_y_disc_bn, dev.get("y_disc_bn", dev["y_disc"])))
        if c["ctc_enable"]:
            pairs += [(self.in_ctc_label, dev["x_ctc_label"]), (self.in_ctc_in_len, dev["x_ctc_in_len"]),
                      (self.in_ctc_out_len, dev["x_ctc_out_len"])]
        # ONE launch for all of them (a table of {destination, source, words}, cached per set of source buffers) when every pair is a
        # plain byte copy; anything else (dtype conversion, strided or unaligned source) goes through torch's copy
        key = tuple((s.data_ptr(), s.dtype, s.numel(), s.is_contiguous()) for _, s in pairs)
        tab = self._copy_tables.get(key) if getattr(self, "_copy_tables_for", None) == (B, T) else None
        if tab is None:
            plain = all(s.is_cuda and s.dtype == d.dtype and s.is_contiguous() and s.numel() == d.numel() and s.data_ptr() % 16 == 0
                        and d.data_ptr() % 16 == 0 for d, s in pairs)
            if getattr(self, "_copy_tables_for", None) != (B, T):
                self._copy_tables = {}; self._copy_tables_for = (B, T)
            if plain and len(self._copy_tables) < 64 and not torch.cuda.is_current_stream_capturing():
                tab = torch.tensor([[d.data_ptr(), s.data_ptr(), d.numel() * d.element_size() // 4] for d, s in pairs],
                                   dtype=torch.int64, device=self.in_x.device)
                self._copy_tables[key] = tab
        if tab is not None:
            _call("sar_copy_ranges", tab, tab.shape[0])
        else:
            for d, s in pairs:
                d.copy_(s.view(d.shape))

    # ------------------------------------------------------------------ forward / backward plans
    def _accent_classifier(self, ar, mode, margin=False):
        """AR_CF_DS1 -> AR_CF_DS2 -> y_accent (+ categorical cross-entropy, + the whole backward down to d(ar) in mode 'train') in one
        launch (csrc/heads_fused.hip): 15 launches of ~1 us of arithmetic otherwise.  The layers keep their parameters, names,
        `.output` tensors and gradient buffers; only who computes them changes.
        margin: the same launch also runs the margin head's training step (forward, cross-entropy, backward; `_fuse_margin`)."""
        l1, l2, l3 = self.AR_CF_DS1, self.AR_CF_DS2, self.y_accent
        B = self.B
        h1, h2, probs = l1.buf("y", (B, l1.units)), l2.buf("y", (B, l2.units)), l3.buf("y", (B, l3.units))
        self._accent_loss = self.store.bufs.get("loss/y_accent/loss", (B,))
        self._accent_dx = l1.buf("dx", (B, l1.in_dim))
        want_loss = mode != "predict" and "y_accent" in self.loss_weights
        train = mode == "train" and want_loss
        args = (ar, l1.kernel.w, l1.bias.w, l2.kernel.w, l2.bias.w, l3.kernel.w, l3.bias.w,
                self.in_y_accent if want_loss else None, self.seed["y_accent"] if train else None, h1, h2, probs, self._accent_loss,
                l1.kernel.g, l1.bias.g, l2.kernel.g, l2.bias.g, l3.kernel.g, l3.bias.g, self._accent_dx)
        if margin:
            d = self.disc
            ws, mprobs, mloss, minvw = d.train_buffers(ar, self.in_x_accent, B)
            _call("sar_accent_margin_train_f32", *args, self.in_x_accent, self.in_y_disc, self.seed["y_disc"], d.W.w, mprobs, mloss, minvw,
                  d.W.g, ws, B, l1.in_dim, l1.units, l2.units, l3.units, d.MODE, d.s, d.m)
            self.out["y_disc"], self._disc_loss = mprobs, mloss
        else:
            _call("sar_accent_classifier_f32", *args, B, l1.in_dim, l1.units, l2.units, l3.units)
        l1.output, l2.output, l3.output = h1, h2, probs
        return probs

    def _fuse_margin(self, mode):
        """The margin head's training step rides in the accent classifier's launch (sar_accent_margin_train_f32) where its
        register-cached form applies: both heads trained, A in {4, 8}, hidden_dim <= 256."""
        c = self.cfg
        return (mode == "train" and self._fused_accent and c["disc_enable"] and c["metric_loss"] in ("sphereface", "cosface", "arcface")
                and "y_disc" in self.loss_weights and "y_accent" in self.loss_weights and c["accent_classes"] in (4, 8)
                and c["hidden_dim"] <= 256 and self.disc.W.w.data_ptr() % 16 == 0 and os.environ.get("SAR_FUSED_HEADS", "1") == "1")

    def _forward(self, mode="predict"):
        """mode: 'predict' (outputs only), 'eval' (+ losses follow), 'train' (+ backward follows) -- what the fused accent classifier
        (forward + loss + backward in one launch) is asked to produce; everything else runs the same in all three."""
        c = self.cfg
        B, L = self.B, self.L
        self._fused_disc = False
        cnn = self.trunk.forward(self.in_x, B)
        x = dense_ln_forward(self.CNN_LIN, self.CNN_LIN_LN, cnn.view(B * L, -1))   # CNN2SEQ (model.py:252) is a view; DS(tanh) -> LN: one launch
        x = self.CRNN.forward(x, B, L)
        crnn = self.CRNN_LN.forward(x.view(B * L, -1))
        self.out = {}
        if c["ctc_enable"]:
            a = self.CTC_BIGRU.forward(crnn, B, L)
            a = self.CTC_BIGRU_LN.forward(a.view(B * L, -1))
            a = dense_ln_forward(self.CTC_DS, self.CTC_DS_LN, a)
            pred = self.ctc_pred.forward(a)
            self.out["ctc_pred"] = pred
            _call("sar_ctc_batch_cost_fwd_f32", pred, self.in_ctc_label, self.in_ctc_in_len, self.in_ctc_out_len, self.ctc_loss,
                  self.ctc_ws, self.ctc_ws_bytes, B, L, c["bpe_classes"], c["max_ctc_len"])
            self.out["y_ctc_loss"] = self.ctc_loss
        if c["ar_enable"]:
            ar = dense_ln_forward(self.AR_DS, self.AR_DS_LN, crnn)
            if c["mto"] == "avg":
                ar = self.merge.forward(ar, B, L)
            else:
                ar = self.merge.forward(ar, B, L)
            ar = self.AR_BN1.forward(ar, B)
            ar = self.AR_EMBEDDING.forward(ar)
            ar = self.AR_BN2.forward(ar, B)
            self.out["ar"] = ar
            both = self._fuse_margin(mode)
            if self._fused_accent:
                self.out["y_accent"] = self._accent_classifier(ar, mode, margin=both)
            else:
                a1 = self.AR_CF_DS2.forward(self.AR_CF_DS1.forward(ar))
                self.out["y_accent"] = self.y_accent.forward(a1)
            if c["disc_enable"]:
                ml = c["metric_loss"]
                if ml == "softmax":
                    self.out["y_disc"] = self.disc.forward(ar)
                elif ml == "circleloss":
                    self.out["y_disc"] = self.disc.forward(self.disc_norm.forward(ar, B))
                elif both:
                    self._fused_disc = True                                   # ran in the classifier's launch above
                elif mode == "train" and self._fused_accent and "y_disc" in self.loss_weights and "y_accent" in self.loss_weights:
                    # margin head: forward + cross-entropy + backward in two launches; its input gradient joins d(ar), which the
                    # classifier's launch above has just written
                    self.out["y_disc"], self._disc_loss = self.disc.train(ar, self.in_x_accent, self.in_y_disc, self.seed["y_disc"],
                                                                         dx_out=self._accent_dx, accumulate_dx=True)
                    self._fused_disc = True
                else:
                    self.out["y_disc"] = self.disc.forward(ar, self.in_x_accent)
                if c["bn_dim"]:
                    v = self.AR_BN3.forward(self.AR_BN_DS.forward(ar), B)
                    v = self.AR_BN4.forward(self.bottleneck.forward(v), B)
                    self.out["bottleneck"] = v
                    if ml == "softmax":
                        self.out["y_disc_bn"] = self.disc_bn.forward(v)
                    elif ml == "circleloss":
                        self.out["y_disc_bn"] = self.disc_bn.forward(self.disc_bn_norm.forward(v, B))
                    else:
                        self.out["y_disc_bn"] = self.disc_bn.forward(v, self.in_x_accent)
        return self.out

    def _losses_forward(self):
        c = self.cfg
        self.loss_vec = {}
        if "y_accent" in self.loss_weights:
            self.loss_vec["y_accent"] = (self._accent_loss if self._fused_accent else
                                         self.cce_accent.forward(self.in_y_accent, self.out["y_accent"]))
        if "y_disc" in self.loss_weights:
            if c["metric_loss"] == "circleloss":
                self.loss_vec["y_disc"] = self.circle.forward(self.in_y_disc, self.out["y_disc"], gamma=256, margin=c["margin"])
            elif self._fused_disc:
                self.loss_vec["y_disc"] = self._disc_loss
            else:
                self.loss_vec["y_disc"] = self.cce_disc.forward(self.in_y_disc, self.out["y_disc"])
        if "y_ctc_loss" in self.loss_weights:
            self.loss_vec["y_ctc_loss"] = self.out["y_ctc_loss"]
        if "y_disc_bn" in self.loss_weights:
            # model.py:364 compares the `metrics` dict with 'circleloss': always unequal, so this output always gets
            # categorical_crossentropy (for circleloss: on the re-normalised raw cosine logits).  Kept verbatim.
            self.loss_vec["y_disc_bn"] = self.cce_disc_bn.forward(self.in_y_disc_bn, self.out["y_disc_bn"])

    def _backward(self):
        self._backward_trunk(self._backward_heads())

    def _backward_heads(self):
        """Backward of everything after the conv trunk (heads, pooling, GRUs, CNN_LIN): completes the tail bucket of the flat
        gradient buffer.  Returns the gradient w.r.t. the trunk output."""
        c = self.cfg
        B, L = self.B, self.L
        d_crnn = None
        self.store.begin_deferred_wgrads()                                   # bf16 GEMM mode: Dense / GRU kernel gradients in one launch, below
        if c["ctc_enable"]:
            _call("sar_ctc_batch_cost_bwd_f32", self.out["ctc_pred"], self.in_ctc_label, self.in_ctc_in_len, self.in_ctc_out_len,
                  self.ctc_loss, self.seed["y_ctc_loss"], self.ctc_ws, self.ctc_ws_bytes, self.ctc_dprobs, B, L, c["bpe_classes"],
                  c["max_ctc_len"])
            g = self.ctc_pred.backward(self.ctc_dprobs)
            g = self.CTC_DS.backward(self.CTC_DS_LN.backward(g, act=self.CTC_DS.activation), dz_ready=True)      # tanh' inside the LayerNorm launch
            g = self.CTC_BIGRU_LN.backward(g)
            d_crnn = self.CTC_BIGRU.backward(g.view(B, L, -1))
        if c["ar_enable"]:
            if self._fused_accent:
                d_ar = self._accent_dx                                        # written by the forward's launch (mode 'train')
            else:
                dp = self.cce_accent.backward(self.seed["y_accent"])
                g = self.y_accent.backward(dp)
                g = self.AR_CF_DS1.backward(self.AR_CF_DS2.backward(g))       # -> d(ar) buffer of AR_CF_DS1
                d_ar = g
            if c["disc_enable"]:
                ml = c["metric_loss"]
                if ml == "circleloss":
                    ds = self.circle.backward(self.seed["y_disc"])
                    dv = self.disc.backward(ds)
                    self.disc_norm.backward(dv, accumulate_dx=True, dx_out=d_ar)
                elif ml == "softmax":
                    dpd = self.cce_disc.backward(self.seed["y_disc"])
                    self.disc.backward(dpd, accumulate_dx=True, dx_out=d_ar)
                elif not self._fused_disc:
                    dpd = self.cce_disc.backward(self.seed["y_disc"])
                    self.disc.backward(dpd, dx_out=d_ar, accumulate_dx=True)
                if c["bn_dim"]:
                    dpb = self.cce_disc_bn.backward(self.seed["y_disc_bn"])
                    if ml == "circleloss":
                        gv = self.disc_bn_norm.backward(self.disc_bn.backward(dpb))
                    else:
                        gv = self.disc_bn.backward(dpb)
                    gv = self.bottleneck.backward(self.AR_BN4.backward(gv))
                    self.AR_BN_DS.backward(self.AR_BN3.backward(gv), dx_out=d_ar, accumulate_dx=True)
            g = self.AR_BN2.backward(d_ar)
            g = self.AR_EMBEDDING.backward(g)
            g = self.AR_BN1.backward(g)
            if c["mto"] == "avg":
                g = self.merge.backward(g)
            elif c["mto"] == "bigru":
                g = self.merge.backward(g)
            else:
                g = self.merge.backward(g)
            g = self.AR_DS_LN.backward(g, act=self.AR_DS.activation)
            if d_crnn is not None:
                self.AR_DS.backward(g, dx_out=d_crnn, accumulate_dx=True, dz_ready=True)
            else:
                d_crnn = self.AR_DS.backward(g, dz_ready=True)
        g = self.CRNN_LN.backward(d_crnn)
        g = self.CRNN.backward(g.view(B, L, -1))
        g = self.CNN_LIN_LN.backward(g, act=self.CNN_LIN.activation)
        g = self.CNN_LIN.backward(g, dz_ready=True)
        self.store.flush_wgrads()
        self._g_trunk = g.view(B, self.trunk.out_hw[0], self.trunk.out_hw[1], -1)
        return self._g_trunk

    def _backward_trunk(self, g):
        """Backward of the ResNet trunk: completes the head bucket (conv stack) of the flat gradient buffer."""
        self.trunk.backward(g)

    def _zero_at_step_start(self):
        # 2-byte trunk: the first launch of the forward pass is the trunk's weight packing, and it takes the zeroing along
        self.store.zero_grads(defer=bool(getattr(self.trunk, "bf16", False)) and os.environ.get("SAR_STEP_BEGIN_MERGE", "1") != "0")

    def _device_step(self, update=True):
        self._zero_at_step_start()
        self._forward("train"); self._losses_forward(); self._backward()
        if update:
            self.store.adam_step()
        elif not self.dp.enabled:
            self.store._zeroed_this_step = False  # (the data-parallel two-graph form runs adam_step as its next segment)

    def _seg_heads(self):
        self._zero_at_step_start()
        self._forward("train"); self._losses_forward(); self._backward_heads()

    def _seg_trunk(self, part):
        self.trunk.backward(self._g_trunk, part=part)

    def _run_segments(self, segments):
        """Run the step as a list of (name, fn, after) segments; `after` (or None) is called on the host between segments
        (collective launches).  First call eager (allocations), second call captures one HIP graph per segment, later calls
        replay them."""
        if self.use_graph and self._eager_done >= 1:
            if self.graph is None:
                torch.cuda.synchronize()
                graphs = {}
                for name, fn, after in segments:
                    g = torch.cuda.CUDAGraph()
                    # thread_local: the feeder thread (utils.Prefetcher) may allocate / copy on its own stream meanwhile
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        fn()
                    graphs[name] = g
                    g.replay()
                    if after:
                        after()
                self.graph = graphs
                return
            for name, fn, after in segments:
                self.graph[name].replay()
                if after:
                    after()
        else:
            for name, fn, after in segments:
                fn()
                if after:
                    after()
            self._eager_done += 1

    # ------------------------------------------------------------------ public step API
    def train_step_device(self):
        """One optimisation step on the batch currently bound in the static input buffers (no host sync).
        world == 1: the whole step is one HIP graph.  world > 1: five graphs -- [forward + heads/GRU backward], [trunk
        backward: stage 3], [stage 2], [stages 1-0 + stem], [L2+Adam] -- and after each backward graph the RCCL all-reduce
        of the gradient bucket it completed is launched asynchronously, so it runs on RCCL's stream underneath the next
        graph; only the last, small bucket (stages 1-0 + stem, 1.5 MB at F=32) is exposed before Adam waits for all."""
        if not self.dp.enabled:
            self._run_segments([("step", self._device_step, None)])
            return
        works = []
        ar = lambda i: (lambda: works.extend(self.dp.allreduce_bucket(i, async_op=True)))
        nparts = self.trunk.N_PARTS
        if len(self.dp.bounds) != nparts + 1:
            # the bucket table does not line up with the backward parts (a split parameter did not resolve): overlapping would
            # reduce ranges that later parts still write.  One all-reduce of everything after the full backward instead.
            def reduce_all():
                for i in range(len(self.dp.bounds)):
                    works.extend(self.dp.allreduce_bucket(i, async_op=True))
                [w.wait() for w in works]
            segs = [("heads", self._seg_heads, None)]
            for part in range(nparts):
                segs.append(("trunk%d" % part, (lambda part=part: self._seg_trunk(part)), reduce_all if part == nparts - 1 else None))
            segs.append(("adam", self.store.adam_step, None))
            self._run_segments(segs)
            return
        seg_mode = os.environ.get("SAR_DP_SEGMENTS") or self._dp_default_segments()
        if seg_mode == "1" and nparts == 3:
            # ONE graph (round 5): the bucket all-reduces are captured INSIDE the step -- RCCL collectives are capturable; the
            # collective's stream forks from the capturing stream where the all-reduce is issued and joins where its handle is waited for, so the
            # replayed graph keeps the overlap of the three-graph form (the 33.6 MB range underneath the backward of stages 1-0 and the stem) without
            # its host gaps (three graph launches, the host-side wait in front of Adam: +7 % on one rank).  Needs a capturable backend: "nccl"
            # (= RCCL) or the library's own sar_allreduce_bucket_f32 (SAR_DP_COLLECTIVE=cabi); gloo reduces through the host and keeps three graphs.
            def step_one():
                self._seg_heads()
                self._seg_trunk("01")
                ws = []
                lo, hi = self.dp.bounds[2][0], self.dp.bounds[0][1]
                if hi > lo and all(self.dp.bounds[i][0] == self.dp.bounds[i + 1][1] for i in (0, 1)):
                    ws.extend(self.dp.allreduce_range(lo, hi, async_op=True))
                else:
                    for i in (0, 1, 2):
                        ws.extend(self.dp.allreduce_bucket(i, async_op=True))
                self._seg_trunk(2)
                ws.extend(self.dp.allreduce_bucket(3, async_op=True))
                [w.wait() for w in ws]
                self.store.adam_step()
            if not getattr(self, "_dp_one_graph_failed", False):
                try:
                    self._run_segments([("step_dp", step_one, None)])
                    return
                except RuntimeError as e:
                    # a collective that cannot be captured on this stack: say so once and run the three-graph form from here on (the failed capture
                    # launched nothing; the eager step in front of it was complete)
                    if self.graph is not None or not (self.use_graph and self._eager_done >= 1):
                        raise
                    import warnings
                    warnings.warn("data-parallel step: capturing the collectives inside the step's HIP graph failed (%s); using three graphs" % (str(e).splitlines()[0],), RuntimeWarning)
                    self._dp_one_graph_failed = True
                    torch.cuda.synchronize()
                    seg_mode = "3"
            else:
                seg_mode = "3"
        if seg_mode == "2":
            # TWO graphs, no overlap: [the whole forward + backward] -> one all-reduce of the flat buffer, waited for -> [L2 + Adam]
            def after_all():
                works.extend(self.dp.allreduce_range(0, self.store.n, async_op=True))
                [w.wait() for w in works]
            self._run_segments([("fwd_bwd", (lambda: self._device_step(update=False)), after_all), ("adam", self.store.adam_step, None)])
            return
        if seg_mode in ("3", "1") and nparts == 3:
            # THREE graphs: [forward + heads / GRU backward + trunk backward down to stage 2] -> [stages 1-0 + stem] -> [L2 + Adam].  The
            # first all-reduce covers the tail, stage-3 and stage-2 buckets -- one contiguous 33.6 MB range of the 35 MB at F=32 -- and
            # runs underneath the second graph (~0.6 ms of backward); only the last 1.5 MB is exposed.  The five-graph form above it
            # (SAR_DP_SEGMENTS=5: one all-reduce per backward part) starts the collectives earlier but pays for it at every N: four host
            # gaps and the weight gradients batched per part -- 0.30 ms on a 2.14 ms step (forced one-rank run), this form 0.1x.
            def seg_a():
                self._seg_heads()
                self._seg_trunk("01")
            def after_a():
                lo, hi = self.dp.bounds[2][0], self.dp.bounds[0][1]
                if hi > lo and all(self.dp.bounds[i][0] == self.dp.bounds[i + 1][1] for i in (0, 1)):
                    works.extend(self.dp.allreduce_range(lo, hi, async_op=True))
                else:
                    for i in (0, 1, 2):
                        works.extend(self.dp.allreduce_bucket(i, async_op=True))
            def after_b():
                works.extend(self.dp.allreduce_bucket(3, async_op=True))
                [w.wait() for w in works]
            self._run_segments([("heads_s32", seg_a, after_a), ("trunk_s10", (lambda: self._seg_trunk(2)), after_b),
                                ("adam", self.store.adam_step, None)])
            return
        segs = [("heads", self._seg_heads, ar(0))]
        for part in range(nparts):
            last = part == nparts - 1
            after = (lambda i=part + 1: (works.extend(self.dp.allreduce_bucket(i, async_op=True)), [w.wait() for w in works])) if last else ar(part + 1)
            segs.append(("trunk%d" % part, (lambda part=part: self._seg_trunk(part)), after))
        segs.append(("adam", self.store.adam_step, None))
        self._run_segments(segs)

    def _dp_default_segments(self):
        """How many HIP graphs the data-parallel step is captured as: "1" where the collectives can be captured with it (backend nccl = RCCL, or
        the C-ABI collective), "3" otherwise (gloo: host-side reduction); SAR_DP_SEGMENTS overrides ("1" | "2" | "3" | "5")."""
        import torch.distributed as dist
        try:
            capturable = self.dp.cabi is not None or dist.get_backend() == "nccl"
        except Exception:
            capturable = False
        return "1" if (capturable and self.use_graph) else "3"

    def train_on_batch(self, x, y=None, return_dict=True, sync=True):
        self._upload(x, y)
        self.train_step_device()
        self._steps_run = getattr(self, "_steps_run", 0) + 1
        if not sync:
            return None
        logs = self._logs()
        self._overflow_policy(logs)
        return logs

    # ---- dynamic loss scaling of the modes with fp16 operands (fp16 trunk; the split-precision mode's fp16 weight-gradient operands and GRU
    # gradients).  The optimiser SKIPS a step whose gradient is not finite (csrc/optim.hip) and counts it on the device; here, wherever the
    # host reads logs anyway (every synchronous train_on_batch, every epoch of fit_generator -- never an extra sync), the counter is
    # looked at: a skipped step is reported (RuntimeWarning + `nonfinite_steps_skipped` in the logs), the gradient scale is halved and the
    # step re-captured; 2 000 clean steps double it again (never past the static 64 x batch); MAX_CONSEC_SKIPS consecutive skipped steps
    # raise -- the reference's Keras trainer would show a NaN loss there.
    MAX_CONSEC_SKIPS = 16
    GROW_AFTER = 2000
    SKIP_POLL_EVERY = 64                                         # fit_generator: steps between two looks at the device's skip counters

    def _overflow_policy(self, logs=None):
        st = self.store
        if not st.__dict__.get("_adam_guard", False):
            return
        run = getattr(self, "_steps_run", 0)
        skipped, in_a_row = st.skip_counters()                   # counted by the optimiser launch itself (csrc/optim.hip: ticket[2], ticket[4])
        new = skipped - getattr(self, "_skips_seen", 0)
        self._skips_seen = skipped
        if logs is not None:
            logs["nonfinite_steps_skipped"] = skipped
        if new > 0:
            import warnings
            self._clean_since = run
            # "in a row" is the DEVICE's word: a host that looks once per epoch / every SKIP_POLL_EVERY steps (fit_generator) cannot tell 16
            # scattered skips from 16 consecutive ones by the total (the old host-side guess raised on any epoch with 16 skips,
            # before the scale was ever lowered).  Every look that finds the run of skips still going on halves the scale; the training has
            # diverged when MAX_CONSEC_SKIPS such looks in a row (= as many halvings) did not end it.
            self._looks_in_run = getattr(self, "_looks_in_run", 0) + 1 if in_a_row > 0 else 0
            if in_a_row >= self.MAX_CONSEC_SKIPS and self._looks_in_run >= self.MAX_CONSEC_SKIPS:
                raise FloatingPointError("%d consecutive optimiser steps skipped: the gradient is not finite at gradient scale 2^%d x (64 x batch) "
                                         "-- the training has diverged (the reference's Keras trainer would report a NaN loss)" % (in_a_row, st.gscale_shift))
            st.gscale_shift -= 1
            self.graph = None; self._eager_done = 0              # the scale is a kernel argument: the next step runs eagerly, the one after re-captures
            warnings.warn("%d optimiser step(s) skipped (%d in all): a gradient element was not finite; gradient scale of the fp16 operands halved "
                          "to 2^%d x (64 x batch)" % (new, skipped, st.gscale_shift), RuntimeWarning)
        else:
            self._looks_in_run = 0
            if st.gscale_shift < 0 and run - getattr(self, "_clean_since", 0) >= self.GROW_AFTER:
                st.gscale_shift += 1; self._clean_since = run
                self.graph = None; self._eager_done = 0

    def _logs(self):
        """Host-side scalars (forces a device sync): Keras-style loss / per-output losses / accuracies."""
        logs = {}
        total = float(self.store.l2_loss.item())
        for k in self.output_names:
            v = float(self.loss_vec[k].mean().item())
            logs[k + "_loss"] = v; total += self.loss_weights[k] * v
        logs["loss"] = total
        for k in self.output_names:
            if k == "y_ctc_loss":
                continue
            tgt = {"y_accent": self.in_y_accent, "y_disc": getattr(self, "in_y_disc", None),
                   "y_disc_bn": getattr(self, "in_y_disc_bn", None)}[k]
            logs[k + "_acc"] = float((self.out[k].argmax(1) == tgt.argmax(1)).float().mean().item())
        return logs

    def _logs_device_names(self):
        return ["loss"] + [k + "_loss" for k in self.output_names] + [k + "_acc" for k in self.output_names if k != "y_ctc_loss"]

    def _logs_device(self):
        """The scalars of _logs() as ONE device vector (order: _logs_device_names()), no host sync: what fit_generator accumulates."""
        per = [self.loss_vec[k].mean() for k in self.output_names]
        total = self.store.l2_loss.reshape(()) + sum(self.loss_weights[k] * v for k, v in zip(self.output_names, per))
        accs = []
        for k in self.output_names:
            if k == "y_ctc_loss":
                continue
            tgt = {"y_accent": self.in_y_accent, "y_disc": getattr(self, "in_y_disc", None),
                   "y_disc_bn": getattr(self, "in_y_disc_bn", None)}[k]
            accs.append((self.out[k].argmax(1) == tgt.argmax(1)).float().mean())
        return torch.stack([total] + per + accs)

    def forward_backward_only(self, x, y=None):
        """For tests: forward + losses + backward without the optimizer update; grads stay in store.grads."""
        self._upload(x, y)
        self.store.zero_grads()
        self._forward("train"); self._losses_forward(); self._backward()
        self.store._zeroed_this_step = False     # no adam_step follows: the next forward must not rely on this step's zeroing
        return self._logs_no_l2()

    def _logs_no_l2(self):
        return {k: self.loss_vec[k].detach().cpu().numpy().copy() for k in self.output_names}

    # ------------------------------------------------------------------ Keras-like conveniences
    def summary(self):
        print('Model: "%s"' % self.name); print("_" * 72)
        print("%-40s %-18s %12s" % ("Layer", "Output shape", "Param #")); print("=" * 72)
        for l in self.store.layers:
            print("%-40s %-18s %12d" % (l.name, str(getattr(l, "output_shape", "")), l.count_params()))
        print("=" * 72); print("Total params: {:,}".format(self.store.n_params()))

    def get_layer(self, name=None):
        for l in self.store.layers:
            if l.name == name:
                return l
        raise ValueError("No such layer: " + str(name))

    def count_params(self):
        return self.store.n_params()

    def get_weights_dict(self):
        return self.store.state_dict()

    def save_weights(self, path):
        np.savez(path if str(path).endswith(".npz") else str(path) + ".npz", **self.store.state_dict())

    def save(self, path):
        """model.save (train.py:35).  No h5py in this image: an .npz of Keras-layout tensors + optimizer state."""
        d = self.store.state_dict()
        d["__adam_m__"] = self.store.m.cpu().numpy(); d["__adam_v__"] = self.store.v.cpu().numpy()
        d["__adam_state__"] = self.store.adam_state.cpu().numpy()
        d["__gscale_shift__"] = np.asarray([self.store.gscale_shift, self.store.skipped_steps()], np.int64)   # dynamic loss scale + skip count
        np.savez(path if str(path).endswith(".npz") else str(path) + ".npz", **d)

    def load_weights(self, path, by_name=True, skip_mismatch=True):
        """model.py:181-183 warm start.  .npz checkpoints of this package, or a Keras .h5 (names mapped by keras_compat; needs
        h5py -- otherwise the error says how to convert the file once)."""
        if str(path).lower().endswith((".h5", ".hdf5")) and os.path.exists(str(path) + ".npz") and not os.path.exists(str(path)):
            # what THIS package's save() wrote for that name (train.py:35 saves '%03d.h5'; without h5py the file is '<name>.h5.npz'): the
            # reference's own save -> raw_model round trip works unchanged
            d = dict(np.load(str(path) + ".npz"))
        elif str(path).lower().endswith((".h5", ".hdf5")):
            from .keras_compat import load_h5
            d = load_h5(path)
        else:
            p = path if str(path).endswith(".npz") else str(path) + ".npz"
            d = dict(np.load(p))
        loaded, skipped = self.store.load_state_dict(d, by_name, skip_mismatch)
        if "__gscale_shift__" in d:
            # a resumed run starts at the gradient scale the saved one had worked its way down to, not at the full 64 x batch it would
            # overflow at again; the scale is a kernel argument, so a captured step is dropped
            shift = min(0, int(np.asarray(d["__gscale_shift__"]).ravel()[0]))
            if shift != self.store.gscale_shift:
                self.store.gscale_shift = shift
                self.graph = None; self._eager_done = 0
        self.dp.broadcast_params()
        return loaded, skipped

    def set_weights_dict(self, d):
        self.store.load_state_dict(d, True, False)

    def fit_generator(self, generator, steps_per_epoch, epochs=1, callbacks=None, initial_epoch=0, validation_data=None,
                      max_queue_size=10, verbose=1, log_every=50, **keras_kwargs):
        """Keras fit_generator as train.py:38-44 uses it (single feeding thread; callbacks get Keras-style logs).  Returns a
        History: `.history` = {log name: [value per epoch]}, `.epoch` = epoch indices; it also iterates / indexes as the list of
        per-epoch log dicts.  Keras keywords that have no meaning here (workers, use_multiprocessing, validation_steps,
        class_weight, shuffle) are accepted and ignored."""
        from .callbacks import CallbackList
        from .utils import Prefetcher
        unknown = set(keras_kwargs) - {"workers", "use_multiprocessing", "validation_steps", "class_weight", "shuffle", "validation_freq"}
        if unknown:
            raise TypeError("fit_generator() got unexpected keyword arguments %s" % sorted(unknown))
        cbs = CallbackList(callbacks or [], self)
        cbs.on_train_begin()
        logs = []
        feed = Prefetcher(generator, max_queue_size)                         # Keras: one feeding thread, bounded queue
        try:
            logs = self._fit_loop(feed, steps_per_epoch, epochs, cbs, initial_epoch, validation_data, verbose, log_every)
        finally:
            feed.close()
        cbs.on_train_end()
        return History(logs, initial_epoch)

    def _fit_loop(self, generator, steps_per_epoch, epochs, cbs, initial_epoch, validation_data, verbose, log_every):
        history = []
        for epoch in range(initial_epoch, epochs):
            cbs.on_epoch_begin(epoch)
            # Keras reports the running mean over EVERY batch of the epoch (CSVLogger / ReduceLROnPlateau / EarlyStopping read it):
            # the per-step scalars are added up on the device (a few tiny launches behind the step's graph, no host sync) and read
            # back once per epoch; `log_every` only paces the optional progress print
            acc = None
            for step in range(steps_per_epoch):
                x, y = next(generator)
                self.train_on_batch(x, y, sync=False)
                vec = self._logs_device()
                acc = vec.clone() if acc is None else acc.add_(vec)
                if verbose > 1 and (step % log_every == 0):
                    print("  step %d/%d loss %.4f" % (step + 1, steps_per_epoch, float(vec[0].item())))
                if (step + 1) % self.SKIP_POLL_EVERY == 0 and step + 1 < steps_per_epoch:
                    self._overflow_policy()                      # one 32-byte read: an overflowing scale is halved within SKIP_POLL_EVERY steps, not at epoch end
            mean_dev = acc / steps_per_epoch
            if self.dp.enabled and self.dp.world > 1:
                # every rank saw its own shard: the callbacks (ReduceLROnPlateau, EarlyStopping, checkpoints on 'loss' / 'acc') must decide
                # on the SAME numbers everywhere, or the replicas take different learning rates / stop in different epochs and the next
                # collective hangs.  One small all-reduce per epoch: the mean over the ranks = the global-batch metrics.
                import torch.distributed as dist
                dist.all_reduce(mean_dev); mean_dev = mean_dev / self.dp.world
            mean = mean_dev.cpu().numpy()
            logs = {k: float(v) for k, v in zip(self._logs_device_names(), mean)}
            self._overflow_policy(logs)                                      # (the .cpu() above was this epoch's sync: the skip counter is current)
            if validation_data is not None:
                vl = self.evaluate(validation_data[0], validation_data[1])
                if self.dp.enabled and self.dp.world > 1:                        # the same validation data on every rank, but be explicit: rank 0's numbers
                    import torch.distributed as dist
                    obj = [vl]; dist.broadcast_object_list(obj, src=0); vl = obj[0]
                logs.update({"val_" + k: v for k, v in vl.items()})
            logs["lr"] = self.store.get_lr()
            if verbose:
                print("Epoch %d/%d - " % (epoch + 1, epochs) + " - ".join("%s: %.4f" % kv for kv in sorted(logs.items())))
            history.append(logs)
            cbs.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        return history

    def _inference(self):
        """Context: Keras learning phase 0 -- BatchNormalization uses its moving averages and does not update them."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev = self.store.training
            self.store.training = False
            try:
                yield
            finally:
                self.store.training = prev
        return ctx()

    def evaluate(self, x, y=None, batch_size=32):
        """Keras `evaluate` (what fit_generator's validation_data runs, train.py:42): inference-mode forward (BN moving
        averages), per-output losses / accuracies and the weighted total incl. the L2 penalties, sample-weighted means
        over all batches (the last one may be smaller)."""
        n = len(x["x_data"]); agg = {}; tot = 0
        with self._inference():
            for s in range(0, n, batch_size):
                sl = slice(s, min(n, s + batch_size))
                xb = {k: v[sl] for k, v in x.items()}; yb = {k: v[sl] for k, v in (y or {}).items()}
                self._upload(xb, yb)
                self._forward("eval"); self._losses_forward()
                self.store.l2_penalty()
                logs = self._logs()
                nb = sl.stop - sl.start; tot += nb
                for k, v in logs.items():
                    agg[k] = agg.get(k, 0.0) + v * nb
        return {k: v / tot for k, v in agg.items()}

    def predict(self, x, batch_size=32):
        """Outputs in the reference's order [y_accent, y_disc, y_ctc_loss] (model.py:327-338), inference mode
        (BN moving averages).  With mode != "train" the time axis may differ from input_shape[0] (model.py:230-231)."""
        x = x if isinstance(x, dict) else {"x_data": x}
        n = len(x["x_data"]); outs = {k: [] for k in self.output_names}
        with self._inference():
            for s in range(0, n, batch_size):
                xb = {k: v[s:s + batch_size] for k, v in x.items()}
                self._upload(xb, None); self._forward()
                for k in self.output_names:
                    outs[k].append(self.out[k].detach().cpu().numpy().copy())
        res = [np.concatenate(outs[k]) for k in self.output_names]
        return res if len(res) > 1 else res[0]


class History(list):
    """keras.callbacks.History as fit_generator returns it (`.history`, `.epoch`), on top of the list of per-epoch log dicts."""

    def __init__(self, logs, initial_epoch=0):
        super().__init__(logs)
        self.epoch = list(range(initial_epoch, initial_epoch + len(logs)))
        self.history = {}
        for l in logs:
            for k, v in l.items():
                self.history.setdefault(k, []).append(v)


def SAR_Net(input_shape, ctc_enable=False, ar_enable=True, disc_enable=False, res_type="res18", res_filters=64,
            hidden_dim=256, bn_dim=0, bpe_classes=1000, accent_classes=8, max_ctc_len=72, mto=None, vlad_clusters=8,
            ghost_clusters=2, metric_loss="cosface", margin=0.3, raw_model=None, lr=0.01, gpus=1, mode="train", name=None,
            seed=0, use_graph=True):
    """model.py:204-371.  Returns (model, train_model); with one process per GPU both are the same object.
    `gpus` > 1 (model.py:193-194, train.py:74): the reference wraps the model in multi_gpu_model inside ONE process; here the
    same train.py runs as `gpus` processes (python -m torch.distributed.run --nproc-per-node <gpus> train.py), this call joins
    the RCCL process group from the launcher's environment, parameters are broadcast from rank 0, and the train_model shards
    every incoming GLOBAL batch itself -- so the script needs no edit beyond the import."""
    if int(gpus or 1) > 1:
        from . import parallel
        _, world, local = parallel.init_from_env()
        if world != int(gpus):
            raise RuntimeError("SAR_Net(gpus=%d) needs one process per GPU: launch the script with `python -m torch.distributed.run "
                               "--nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 <script>` (WORLD_SIZE is %d)" % (gpus, gpus, world))
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
    cfg = dict(locals())
    model = SARNetModel(cfg)
    model.summary()                                                         # model.py:180
    if raw_model:
        print("===== init weights from:%s =====" % raw_model)             # model.py:181-183
        model.load_weights(raw_model, by_name=True, skip_mismatch=True)
    return model, model


class SubModel:
    """What `Model(inputs=model.get_layer(input_name).input, outputs=model.get_layer(output_name).output)` gives the caller of
    the reference's sub_model (model.py:380-383): an object whose predict() returns the named layer's output -- e.g. the
    utterance embedding `AR_BN2`, the pooled vector `gvlad_pool`, or the encoder states `CRNN_LN` -- in inference mode."""

    def __init__(self, model, input_name, output_name):
        if input_name != "x_data":
            raise ValueError("sub-models start at the spectrogram input 'x_data' (got %r)" % (input_name,))
        self.model, self.layer = model, model.get_layer(output_name)
        self.name = output_name

    def predict(self, x, batch_size=32):
        m = self.model
        x = x if isinstance(x, dict) else {"x_data": x}
        n = len(x["x_data"]); outs = []
        with m._inference():
            for s in range(0, n, batch_size):
                xb = {k: v[s:s + batch_size] for k, v in x.items()}
                m._upload(xb, None); m._forward()
                o = self.layer.output
                B = m.B
                if o.dim() == 2 and o.shape[0] == B * m.L and m.L > 1:       # sequence tensors are (B*L, C) on the device
                    o = o.view(B, m.L, -1)
                outs.append(o.detach().cpu().numpy().copy())
        return np.concatenate(outs)


def sub_model(model, input_name, output_name):
    """model.py:380-383."""
    return SubModel(model, input_name, output_name)


def ctc_pred(model, x, batch_size, input_len):
    """model.py:385-389: greedy CTC decode of the ctc_pred probabilities."""
    x = x if isinstance(x, dict) else {"x_data": x}
    res = []
    n = len(x["x_data"])
    with model._inference():                                                  # model.predict: Keras learning phase 0
        for s in range(0, n, batch_size):
            xb = {k: v[s:s + batch_size] for k, v in x.items()}
            model._upload(xb, None); model._forward()
            B, L, V = model.B, model.L, model.cfg["bpe_classes"]
            il = torch.full((B,), int(input_len), dtype=torch.int32, device="cuda")
            dec = torch.empty(B, L, dtype=torch.int32, device="cuda"); dl = torch.empty(B, dtype=torch.int32, device="cuda")
            _call("sar_ctc_greedy_decode_f32", model.out["ctc_pred"], il, dec, dl, B, L, V)
            dec, dl = dec.cpu().numpy(), dl.cpu().numpy()
            res += [dec[b, :dl[b]] for b in range(B)]
    # K.get_value(K.ctc_decode(...)[0][0]) (model.py:385-389): one dense int array, rows padded with -1 to the longest decoding
    width = max([len(r) for r in res] + [0])
    out = np.full((len(res), width), -1, dtype=np.int64)
    for i, r in enumerate(res):
        out[i, :len(r)] = r
    return out
