"""Host-side runtime for the SAR-Net hot path: flat parameter/gradient/Adam buffers in HBM, a scratch pool,
and thin wrappers that launch the C-ABI kernels (include/sarnet.h) on torch's current HIP stream.

PyTorch is used for device memory, streams, HIP-graph capture and torch.distributed only; every FLOP of the
train step runs in libsarnet_hip.so.  There is no CPU path: constructing a ParamStore without a HIP device
raises.
"""
import math
import os
import numpy as np
import torch

from . import _lib

CHUNK = 256            # Adam/L2 chunk (elements); every parameter is padded to a multiple of it
ACT = {None: 0, "linear": 0, "tanh": 1, "relu": 2}


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("aesrc2020_amd needs a HIP device (MI355X); there is no CPU fallback. "
                           "The CPU oracle lives in oracle/ and is test infrastructure only.")
    _lib.lib()


class ParamRef:
    __slots__ = ("name", "shape", "init", "l2", "offset", "size", "w", "g", "constraint")

    def __init__(self, name, shape, init, l2, constraint=None):
        self.name, self.shape, self.init, self.l2, self.constraint = name, tuple(shape), init, float(l2), constraint
        self.size = int(np.prod(shape)); self.offset = -1; self.w = None; self.g = None


class ParamStore:
    """Flat fp32 buffers [params | grads | adam m | adam v] + per-chunk L2 table.
    Parameters are laid out in creation (= forward) order, so the tail of the gradient buffer (heads, GRUs)
    is complete first during backward and the head (conv stack) last -- the two all-reduce buckets of parallel.py."""

    def __init__(self, seed=0, device="cuda"):
        self.refs = []; self.by_name = {}; self.finalized = False
        self.state_refs = []          # non-trainable (BN moving statistics)
        self.rng = np.random.default_rng(seed)
        self.bufs = Buffers(device)
        self.layers = []              # construction order; drives summary()/get_layer()
        self.training = True          # Keras learning phase: False = BatchNorm uses its moving averages (predict/evaluate)
        # power-of-two exponent (<= 0) applied on top of the static gradient scale of the fp16 operands (64 x batch): lowered by the model when
        # a step overflowed, raised again after 2 000 clean steps (dynamic loss scaling at the cadence of the host's log reads: no per-step
        # sync).  Per store (as a class attribute models shared it until the first write) and part of save() / load_weights().
        self.gscale_shift = 0

    def register_layer(self, layer):
        self.layers.append(layer)

    def auto_name(self, base):
        """Keras-style auto names (conv2d_1, batch_normalization_1, ...) in construction order."""
        self._auto = getattr(self, "_auto", {})
        self._auto[base] = self._auto.get(base, 0) + 1
        return "%s_%d" % (base, self._auto[base])

    def add(self, name, shape, init, l2=0.0, constraint=None):
        assert not self.finalized, "ParamStore already finalized"
        if name in self.by_name:
            raise ValueError("duplicate parameter name " + name)
        r = ParamRef(name, shape, np.asarray(init, np.float32).reshape(shape), l2, constraint)
        self.refs.append(r); self.by_name[name] = r
        return r

    def add_state(self, name, shape, init):
        r = ParamRef(name, shape, np.asarray(init, np.float32).reshape(shape), 0.0)
        self.state_refs.append(r); self.by_name[name] = r
        return r

    def finalize(self, device="cuda"):
        require_gpu()
        off = 0
        for r in self.refs:
            r.offset = off; off += -(-r.size // CHUNK) * CHUNK
        self.n = off
        host = np.zeros(off, np.float32); l2 = np.zeros(off // CHUNK, np.float32)
        for r in self.refs:
            host[r.offset:r.offset + r.size] = r.init.ravel()
            l2[r.offset // CHUNK:(r.offset + r.size + CHUNK - 1) // CHUNK] = r.l2
        self.flat = torch.zeros(4, off, dtype=torch.float32, device=device)
        self.params, self.grads, self.m, self.v = self.flat[0], self.flat[1], self.flat[2], self.flat[3]
        self.params.copy_(torch.from_numpy(host))
        self.chunk_l2 = torch.from_numpy(l2).to(device)
        for r in self.refs:
            r.w = self.params[r.offset:r.offset + r.size].view(r.shape)
            r.g = self.grads[r.offset:r.offset + r.size].view(r.shape)
            r.init = None
        for r in self.state_refs:
            r.w = torch.from_numpy(r.init).to(device); r.init = None
        # device-side Adam state block: {lr, decay, b1, b2, eps, lr_t, int iterations, grad_scale}
        self.adam_state = torch.zeros(8, dtype=torch.float32, device=device)
        self.l2_loss = torch.zeros(1, dtype=torch.float32, device=device)
        self.finalized = True
        return self

    def set_adam(self, lr, decay=2e-4, b1=0.9, b2=0.999, eps=1e-7, iterations=0, grad_scale=1.0):
        h = torch.tensor([lr, decay, b1, b2, eps, 0.0, 0.0, grad_scale], dtype=torch.float32)
        h.view(torch.int32)[6] = int(iterations)
        self.adam_state.copy_(h)

    def set_lr(self, lr):
        self.adam_state[0:1].copy_(torch.tensor([lr], dtype=torch.float32))

    def get_lr(self):
        return float(self.adam_state[0].item())

    @property
    def iterations(self):
        return int(self.adam_state.view(torch.int32)[6].item())

    # ---- buffers that must be zero when a train step starts (gradients, the penalty scalar, and whatever layers register: BatchNorm
    # statistic bins, split-K outputs): ONE launch at the top of the step instead of one fill launch each
    def zero_at_step_start(self, t):
        """Called by a layer where it would zero `t` itself.  True: the step-start launch of THIS step already covered `t` (skip the
        fill); False: zero it here (first sighting of the buffer, or no step-start launch ran: evaluate / predict)."""
        key = (t.data_ptr(), t.numel())
        zl = self.__dict__.setdefault("_zero_list", {})
        if key not in zl:
            zl[key] = t
            return False
        # covered only if the table the step-start launch used already had this buffer's row
        return self.__dict__.get("_zeroed_this_step", False) and key in self.__dict__.get("_zero_rows", ())

    def _zero_table_sync(self):
        """(Re)build the device table of the zero list; never inside a HIP-graph capture (it is a host-to-device copy): the eager first
        step registers every buffer and rebuilds the table in its adam_step, so the captured step finds it ready."""
        zl = self.__dict__.setdefault("_zero_list", {})
        for t in (self.grads, self.l2_loss):
            zl.setdefault((t.data_ptr(), t.numel()), t)
        tab = self.__dict__.get("_zero_table")
        if (tab is None or tab.shape[0] != len(zl)) and not torch.cuda.is_current_stream_capturing():
            self._zero_table = torch.tensor([[t.data_ptr(), t.numel() * t.element_size() // 4] for t in zl.values()], dtype=torch.int64, device=self.grads.device)
            self._zero_rows = set(zl)
        return self.__dict__.get("_zero_table")

    # ---- deferred weight gradients: the Dense / GRU kernel gradients of a backward pass are off its critical path; in the bf16 GEMM
    # mode the layers hand them over here and ONE grouped launch (csrc/gemm.hip: gemm_bf16_wgrad_group_kernel) runs them behind the
    # last input gradient -- 5 (cfg2) to 10 (cfg3) launches of 7-35 us that each filled a fraction of the chip
    def begin_deferred_wgrads(self):
        self._defer = [] if (self.training and _lib.get_gemm_math() in ("bf16", "bf16x3") and os.environ.get("SAR_WGRAD_GROUP", "1") == "1") else None

    PER_UTT_SHIFT = 6        # gradient rows per utterance, not per frame: ~L times larger, times rstd <= 31 behind a BatchNorm -- 2^-6 of the scale

    def bwd_gemm_shift(self, rows, per_utt=None):
        """per_utt: the layer's gradient rows are per utterance (one row per batch element: AR_EMBEDDING and what follows it).  A layer that does
        not say is classed by its row count: rows <= batch is one row per utterance -- a per-frame layer has batch x L rows, and at L = 1 the
        two are the same thing (round 5 tested rows < 8 x batch, which gave per-frame layers of short inputs the extra 2^-6: ADVICE r5)."""
        if per_utt is None:
            per_utt = rows <= max(self.__dict__.get("batch", 64), 1)
        return self.PER_UTT_SHIFT if per_utt else 0

    def bwd_gemm_scale(self, rows=None):
        """> 0: the backward products of the per-frame Dense / GRU-projection layers run single-pass on fp16 operands, the gradient operand
        times this power of two (csrc/gemm.hip: F16) -- the split-precision GEMM mode's backward since round 5 (SAR_GEMM_BWD=x3: three
        passes as in the forward).  The scale is the 2-byte trunk's: 64 x batch, lowered by the overflow guard (gscale_shift)."""
        if not self.training or _lib.get_gemm_math() != "bf16x3" or os.environ.get("SAR_GEMM_BWD", "f16") != "f16":
            return 0.0
        if not self.__dict__.get("_adam_guard", False):                      # no overflow guard on this store: no fp16 gradient operands
            return 0.0
        shift = self.bwd_gemm_shift(rows) if rows is not None else 0
        return float(2 ** (int(round(np.log2(64.0 * max(self.__dict__.get("batch", 64), 1)))) + self.gscale_shift - shift))

    def deferring_wgrads(self):
        return self.__dict__.get("_defer") is not None

    def defer_wgrad(self, X, dY, dW, db, in_dim, out_dim, R, lda, ldb, ldc, splitk):
        """The operands must stay as they are until flush_wgrads(): per-layer buffers, not shared scratch."""
        # split-K: pick_splitk aims a SINGLE launch at 256 blocks; the group fills the chip anyway and every extra slice is another
        # round of atomics (tools/gemm_group_bench.py, cfg2 problem set: 1 / 2 / 4 / 8 slices = 85 / 60 / 54 / 67 us)
        self._defer.append((X, dY, dW, db, in_dim, out_dim, R, lda, ldb, ldc, min(splitk, 4), self.bwd_gemm_shift(R)))

    def flush_wgrads(self):
        probs, self._defer = self.__dict__.get("_defer"), None
        if not probs:
            return
        key = tuple((q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr()) + tuple(q[4:]) for q in probs)
        groups = self.__dict__.setdefault("_wg_groups", {})
        grp = groups.get(key)
        if grp is None and not torch.cuda.is_current_stream_capturing():      # (the table upload cannot be captured: built in the eager step)
            def eligible(q):
                try:
                    _lib.GemmWgradGroup([q]); return True
                except ValueError:
                    return False
            flags = [eligible(q) for q in probs]                                # the rest (a shape off the bf16 path): one launch each
            if sum(flags) < 2:
                flags = [False] * len(probs)
            ok = [q for q, f in zip(probs, flags) if f]
            grp = (_lib.GemmWgradGroup(ok) if ok else None, [q for q, f in zip(probs, flags) if not f])
            if len(groups) > 16:
                groups.clear()
            groups[key] = grp
        if grp is None:                                                         # first sighting inside a capture: as without deferral
            grp = (None, probs)
        if grp[0] is not None:
            grp[0].launch(self.bwd_gemm_scale())
        for q in grp[1]:
            _lib.call("sar_gemm_wgrad_f32", *q[:11])

    def zero_grads(self, defer=False):
        """defer: the caller promises that the table this returns is zeroed by the very next launch (the 2-byte trunk's weight packing takes
        it along: sar_conv_pack_weights_zero_*) -- one launch fewer at the start of every step."""
        tab = self._zero_table_sync()
        self._zeroed_this_step = True
        if defer:
            self._deferred_zero = tab
            return tab
        _lib.call("sar_zero_ranges", tab, tab.shape[0])
        return None

    def set_adam_guard(self, on=True):
        """Modes with fp16 operands (the fp16 trunk; the split-precision mode's fp16 weight-gradient operands and GRU gradients): a step whose
        gradient holds a non-finite element (a scaled value past fp16's range) is SKIPPED as a whole by the optimiser -- weights, m, v and
        the iteration count stay -- and counted; skipped_steps() reads the counter, the model halves the gradient scale when it moves."""
        self._adam_guard = bool(on)
        if self.__dict__.get("_adam_ticket") is not None:
            self._adam_ticket[1] = 1 if on else 0

    def skipped_steps(self):
        t = self.__dict__.get("_adam_ticket")
        return 0 if t is None else int(t[2].item())

    def skip_counters(self):
        """(steps skipped in all, steps skipped IN A ROW up to the last one run): one 8-word read of the optimiser's ticket block."""
        t = self.__dict__.get("_adam_ticket")
        if t is None:
            return 0, 0
        h = t.cpu()
        return int(h[2]), int(h[4])

    dropped_gradient_elements = skipped_steps                                # (round 3's name, kept for callers)

    def take_deferred_zero(self):
        tab = self.__dict__.get("_deferred_zero"); self._deferred_zero = None
        return tab

    def adam_step(self):
        """L2 gradient + Keras Adam over the whole flat buffer, then the unit_norm constraints."""
        if not self.__dict__.get("_zeroed_this_step", False):
            _lib.call("sar_fill_f32", self.l2_loss, 0.0, 1)
        self._zeroed_this_step = False                                       # the step is over: later forwards (evaluate) zero for themselves
        self._zero_table_sync()                                              # buffers registered during this (eager) step join the table now
        if self.__dict__.get("_adam_ticket") is None:
            self._adam_ticket = torch.zeros(8, dtype=torch.int32, device=self.params.device)      # [ticket, guard switch, skipped, flag, skipped in a row, spare x3]
            if self.__dict__.get("_adam_guard", False):
                self._adam_ticket[1] = 1
        if self.__dict__.get("_adam_guard", False):                          # any non-finite gradient element -> the update below skips the step
            _lib.call("sar_grads_check_f32", self.grads, self._adam_ticket, self.n)
        _lib.call("sar_adam_l2_update_f32", self.params, self.grads, self.m, self.v, self.chunk_l2, self.adam_state, self._adam_ticket,
                  self.l2_loss, self.n)                                      # step-size bookkeeping + update in one launch
        for r in self.refs:
            if r.constraint == "unit_norm":
                _lib.call("sar_unit_norm_f32", r.w, r.shape[0], r.shape[1])

    def l2_penalty(self):
        """Recompute l2_loss for the current weights without an update (evaluate())."""
        _lib.call("sar_l2_penalty_f32", self.params, self.chunk_l2, self.l2_loss, self.n)

    def n_params(self):
        return sum(r.size for r in self.refs)

    def state_dict(self):
        """name -> numpy array in the reference's (Keras) tensor layouts."""
        d = {r.name: r.w.detach().cpu().numpy().copy() for r in self.refs}
        d.update({r.name: r.w.detach().cpu().numpy().copy() for r in self.state_refs})
        return d

    def load_state_dict(self, d, by_name=True, skip_mismatch=True):
        loaded, skipped = [], []
        for r in self.refs + self.state_refs:
            if r.name not in d:
                continue
            a = np.asarray(d[r.name], np.float32)
            if tuple(a.shape) != r.shape:
                if skip_mismatch:
                    skipped.append(r.name); continue
                raise ValueError("shape mismatch for %s: %s vs %s" % (r.name, a.shape, r.shape))
            r.w.copy_(torch.from_numpy(a).to(r.w.device)); loaded.append(r.name)
        return loaded, skipped


# ------------------------------------------------------------------------------------------------ store context
_STACK = []


class store_scope:
    """`with store_scope(store):` -- layers constructed inside register their weights there (the analogue of the
    Keras default graph the reference's layer constructors rely on)."""

    def __init__(self, store):
        self.store = store

    def __enter__(self):
        _STACK.append(self.store); return self.store

    def __exit__(self, *a):
        _STACK.pop()


def current_store():
    if not _STACK:
        raise RuntimeError("no active ParamStore: construct layers inside `with store_scope(store):` (SAR_Net does)")
    return _STACK[-1]


# ------------------------------------------------------------------------------------------------ buffers
class Buffers:
    """Named static device buffers (activations, saved state, scratch).  Shapes are fixed per model instance, so
    everything is allocated once (first eager step) and then reused -- a requirement for HIP-graph replay."""

    def __init__(self, device="cuda"):
        self.device = device; self.d = {}

    def get(self, key, shape, dtype=torch.float32):
        shape = tuple(int(s) for s in shape)
        t = self.d.get(key)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device); self.d[key] = t
        return t

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.d.values())


# ------------------------------------------------------------------------------------------------ initialisers
def he_normal(rng, shape, fan_in):
    """keras he_normal = VarianceScaling(2, fan_in, truncated normal) (model.py:39; resnet.py:34,54,86,116)."""
    std = math.sqrt(2.0 / fan_in) / .87962566103423978
    w = rng.standard_normal(shape) * std
    bad = np.abs(w) > 2 * std
    while bad.any():
        w[bad] = rng.standard_normal(int(bad.sum())) * std; bad = np.abs(w) > 2 * std
    return w


def glorot_uniform(rng, shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, shape)


def orthogonal(rng, shape):
    """keras Orthogonal(gain=1): (VLAD.py:19; model.py:91,101)."""
    rows, cols = shape[0], int(np.prod(shape[1:]))
    a = rng.standard_normal((rows, cols))
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    return (u if u.shape == (rows, cols) else vt).reshape(shape)


# ------------------------------------------------------------------------------------------------ kernel wrappers
def same_pads(n, k, s):
    out = -(-n // s); tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2


def gemm(A, B, C, bias, M, N, K, lda, ldb, ldc, a_kc, b_kc, act=0, accumulate=0, splitk=1):
    _lib.call("sar_gemm_f32", A, B, C, bias, M, N, K, lda, ldb, ldc, a_kc, b_kc, act, accumulate, splitk)


def pick_splitk(M, N, K):
    tiles = -(-M // 64) * -(-N // 64)
    if tiles >= 128 or K < 256:
        return 1
    if K < 512:                                                    # a skinny output (the 34-column assignment layer: 39 tiles on 256 CUs)
        return int(max(1, min(4, 256 // tiles, K // 64))) if tiles < 64 else 1
    return int(max(1, min(32, 256 // tiles, K // (128 if tiles <= 8 else 256))))   # a handful of tiles (the assignment layer's weight gradient): finer slices
