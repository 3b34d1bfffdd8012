"""ctypes binding of libsarnet_hip.so (the C-ABI declared in include/sarnet.h).

The prototypes are parsed from the header so it stays the single source of truth.  There is NO fallback:
if the shared library is missing or a symbol is absent, importing/calling fails loudly.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAR_LIB_PATH") or os.path.join(_HERE, "libsarnet_hip.so")   # override: developer A/B builds only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sarnet.h")

_CT = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "size_t": ctypes.c_size_t}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every `int|size_t sar_*(...)` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|size_t)\s+(sar_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        at = []
        for a in args.split(","):
            a = a.strip()
            if a in ("void", ""):
                continue
            if "*" in a or "sarStream_t" in a:
                at.append(ctypes.c_void_p)
            else:
                at.append(_CT[a.split()[-2] if len(a.split()) > 1 else a])
        protos[name] = (_CT[ret], at)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `python __graft_entry__.py build` "
                              "(make -C aesrc2020_amd/csrc). There is no CPU fallback." % LIB_PATH)
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.fn = {}
        for name, (ret, at) in self.protos.items():
            try:
                f = getattr(self.cdll, name)      # AttributeError if the .so lacks a declared symbol
            except AttributeError:
                if os.environ.get("SAR_LIB_PATH"):                        # developer A/B run against an OLDER build of the library (tools/ab_bench.sh):
                    continue                                              # entry points added since are absent there; calling one still fails loudly
                raise
            f.restype = ret; f.argtypes = at
            self.fn[name] = f
        # The library's own default for the Dense / GRU-projection GEMMs is fp32 MFMA (sar_set_gemm_math(0)); the PACKAGE's default is the
        # fp32-class fast mode, like the trunk's (DEFAULT_MODE below): split precision, unless SAR_GEMM_MATH says otherwise.
        if "SAR_GEMM_MATH" not in os.environ:
            self.fn["sar_set_gemm_math"](GEMM_MATH[DEFAULT_MODE[1]])


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def ptr(t):
    """torch tensor (or None) -> raw device pointer."""
    return None if t is None else t.data_ptr()


# ---- plan workspace of the 2-byte trunk's conv launcher (csrc/conv_bf16.hip: conv_plan_kernel; include/sarnet.h: sar_conv_plan_workspace_*).
# The library never allocates: the package lends each build of the launcher ("bf16" | "f16" | "x3") one buffer for the life of the process --
# captured graphs keep pointers into it, so it is never freed or handed out again -- the first time a conv of that build is launched.
_PLAN_WS = {}
_PLAN_SFX = {"sar_conv2d_bf16": "bf16", "sar_conv2d_f16": "f16", "sar_conv2d_f16a": "f16", "sar_conv2d_x3": "x3",
             "sar_conv2d_pair_bf16": "bf16", "sar_conv2d_pair_f16": "f16", "sar_conv2d_pair_f16a": "f16", "sar_conv2d_pair_x3": "x3"}


def conv_plan_workspace(sfx):
    """Make sure build `sfx` of the conv launcher has its plan workspace (SAR_CONV_PLAN_MB megabytes, default 32; 0: none -- every block of
    every launch then derives its tables itself, as before round 6).  Not inside a graph capture (it allocates)."""
    import torch
    if sfx in _PLAN_WS or torch.cuda.is_current_stream_capturing():
        return
    mb = int(os.environ.get("SAR_CONV_PLAN_MB", "32"))
    if mb <= 0 or "sar_conv_plan_workspace_" + sfx not in lib().fn:          # (absent: an older build under SAR_LIB_PATH)
        _PLAN_WS[sfx] = None
        return
    buf = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")            # on the process's current device: one process per GPU (parallel.py)
    if lib().fn["sar_conv_plan_workspace_" + sfx](buf.data_ptr(), buf.numel()) != 0:
        raise RuntimeError("sar_conv_plan_workspace_%s failed" % sfx)
    _PLAN_WS[sfx] = buf


def call(name, *args, stream=None):
    """Invoke a C-ABI entry point on torch's current HIP stream; raises on a non-zero status."""
    import torch
    L = lib()
    if name in _PLAN_SFX and _PLAN_SFX[name] not in _PLAN_WS:
        conv_plan_workspace(_PLAN_SFX[name])
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    conv = [a.data_ptr() if hasattr(a, "data_ptr") else (ctypes.byref(a) if isinstance(a, ctypes.Structure) else a) for a in args]
    rc = L.fn[name](*conv, stream)
    if rc != 0:
        raise RuntimeError("%s failed with status %d (args=%r)" % (name, rc, [getattr(a, "shape", a) for a in args]))


CONV_MATH = {"fp32": 0, "bf16x3": 1}


def set_conv_math(mode):
    """'fp32' (exact fp32 MFMA) or 'bf16x3' (split-precision, default) for the wide conv layers; host-side, process-wide."""
    rc = lib().fn["sar_set_conv_math"](CONV_MATH[mode])
    if rc != 0:
        raise RuntimeError("sar_set_conv_math failed")


def get_conv_math():
    return {v: k for k, v in CONV_MATH.items()}[lib().fn["sar_get_conv_math"]()]


GEMM_MATH = {"fp32": 0, "bf16": 1, "bf16x3": 2}


def set_gemm_math(mode):
    """'fp32' (default), 'bf16' (operands rounded to bf16 in the staging, bf16 MFMA, fp32 accumulate) or 'bf16x3' (operands split into
    hi / lo bf16 planes in the staging, three MFMAs per product block: fp32-class) for the Dense / GRU projection GEMMs whose extents
    are multiples of 4 from 64 up; 'bf16' / 'bf16x3' also put the hidden_dim-256 GRU recurrence on the MFMA kernels (bf16 / fp16
    operands); host-side, process-wide."""
    if lib().fn["sar_set_gemm_math"](GEMM_MATH[mode]) != 0:
        raise RuntimeError("sar_set_gemm_math failed")


def get_gemm_math():
    return {v: k for k, v in GEMM_MATH.items()}[lib().fn["sar_get_gemm_math"]()]


# ---- activation storage of the conv trunk: "fp32" (round-1 kernels, conv math per set_conv_math), "bf16" / "fp16" (conv_bf16.hip:
# 2-byte activations / activation gradients, single-pass MFMA, fp32 statistics and weights) or "bf16x3" (the same kernels in split
# precision, csrc/trunk16.h SAR_X3: fp32 trunk tensors in HBM, every conv operand a (hi, lo) pair of bf16 planes in LDS, three MFMAs per product block -- the
# fp32-class mode that holds north_star's 1e-3).  Read when a model is built.
# The package default (round 4): "bf16x3" for the trunk AND the GEMMs -- what a SAR_Net(...) built by the reference's train.py runs in: outputs
# within north_star's 1e-3 of the fp32 graph (measured ~1e-4) at 0.6x the step time of the fp32-storage kernels.  Widths the 2-byte kernels
# do not take (res_filters != 32) fall back to those kernels inside ResNetTrunk.  SAR_TRUNK_DTYPE / SAR_GEMM_MATH override.
DEFAULT_MODE = ("bf16x3", "bf16x3")
_TRUNK_DTYPE = [os.environ.get("SAR_TRUNK_DTYPE", DEFAULT_MODE[0])]
TRUNK_DTYPES = ("fp32", "bf16", "fp16", "bf16x3")


def set_trunk_dtype(dtype):
    if dtype not in TRUNK_DTYPES:
        raise ValueError("trunk dtype must be one of %r, got %r" % (TRUNK_DTYPES, dtype))
    _TRUNK_DTYPE[0] = dtype


# fp16 trunk: static scale S of the activation gradients inside the trunk (include/sarnet.h, "fp16 trunk").  Power of two, so scaling
# and unscaling are exact.  0 = automatic: S = 64 x batch (rounded to a power of two) -- the losses are batch means, so gradient
# magnitudes go with 1/B; measured with tools/numerics_table.py on the parity and bench configurations: the largest scaled gradient
# stays 10-100x below fp16's 65504 and < 0.01 % of the non-zero values fall below the smallest normal.  Read when a model is built.
_TRUNK_GSCALE = [float(os.environ.get("SAR_F16_GSCALE", 0.0))]


def set_trunk_gscale(s):
    _TRUNK_GSCALE[0] = float(s)


# Deterministic forward statistics of the 2-byte trunk (csrc/trunk16.h: fix28): BatchNorm sums as 64-bit fixed point, integer atomics --
# independent of block / atomic order and of how the batch is cut into tiles.  Read when a model is built.
_BN_DET = [os.environ.get("SAR_BN_DET", "0") == "1"]


def set_bn_deterministic(on):
    _BN_DET[0] = bool(on)


def get_bn_deterministic():
    return _BN_DET[0]


def get_trunk_gscale():
    return _TRUNK_GSCALE[0]


def get_trunk_dtype():
    return _TRUNK_DTYPE[0]


class BnRef(ctypes.Structure):
    """include/sarnet.h: sarBnRef."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("gamma", "beta", "bins", "bbins", "saved", "moving_mean", "moving_var", "dgamma", "dbeta")] + \
               [(n, ctypes.c_float) for n in ("inv_n", "unbias", "eps", "momentum")] + [("write", ctypes.c_int), ("ginv", ctypes.c_float), ("det", ctypes.c_int), ("pad_", ctypes.c_int)]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        if "ginv" not in k and len(a) < 15:
            self.ginv = 1.0                                                  # no gradient scale unless the caller sets one


class ConvArgs(ctypes.Structure):
    """include/sarnet.h: sarConvArgs -- the arguments of one sar_conv2d_* call (sar_conv2d_pair_* takes two)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("src", "src2", "src3", "wpk", "bias", "res", "y", "side")] + \
               [(n, ctypes.c_int) for n in ("B", "Hs", "Ws", "Cin", "Ho", "Wo", "Cout", "K", "s", "d", "pt", "pl", "fill")] + \
               [("fbn", ctypes.c_void_p), ("epi", ctypes.c_int), ("e_bins", ctypes.c_void_p), ("e_y", ctypes.c_void_p), ("e_saved", ctypes.c_void_p)] + \
               [("side2", ctypes.c_void_p), ("side_scale", ctypes.c_float)]           # sar_conv2d_x3's two extra arguments (absent: NULL / 0)

    @classmethod
    def of(cls, *args):
        """From the positional arguments of sar_conv2d_* (tensors, None, ints, a BnRef or None) -- keeps what it points to alive."""
        a = cls()
        a.keep = [x for x in args if hasattr(x, "data_ptr") or isinstance(x, ctypes.Structure)]
        for (name, _), v in zip(cls._fields_, args):
            if hasattr(v, "data_ptr"):
                v = v.data_ptr()
            elif isinstance(v, ctypes.Structure):
                v = ctypes.addressof(v)
            setattr(a, name, v)
        return a


def conv_pair(a, b, sfx="bf16"):
    """sar_conv2d_pair_*: both convolutions in one launch -> True; False (nothing launched) when their instantiations do not pair up."""
    import torch
    if _PLAN_SFX["sar_conv2d_pair_" + sfx] not in _PLAN_WS:
        conv_plan_workspace(_PLAN_SFX["sar_conv2d_pair_" + sfx])
    rc = lib().fn["sar_conv2d_pair_" + sfx](ctypes.byref(a), ctypes.byref(b), torch.cuda.current_stream().cuda_stream)
    if rc == -1:
        return False
    if rc != 0:
        raise RuntimeError("sar_conv2d_pair_%s failed with status %d" % (sfx, rc))
    return True


class GemmBatch(ctypes.Structure):
    """include/sarnet.h: sarGemmBatch -- up to four (A, B, C, bias) pointer sets of equal-shape GEMMs for one launch."""
    _fields_ = [(n, ctypes.c_void_p * 4) for n in ("A", "B", "C", "bias")]

    def __init__(self, sets):
        super().__init__()
        assert 1 <= len(sets) <= 4
        self.n = len(sets)
        self.keep = sets                                                     # the tensors whose addresses the struct holds
        for i, st in enumerate(sets):
            for name, t in zip(("A", "B", "C", "bias"), st):
                getattr(self, name)[i] = None if t is None else t.data_ptr()


class GemmWgrad(ctypes.Structure):
    """include/sarnet.h: sarGemmWgrad."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("X", "dY", "dW", "db")] + \
               [(n, ctypes.c_int) for n in ("in_dim", "out_dim", "R", "lda", "ldb", "ldc", "splitk", "pad_")]


class GemmWgradGroup:
    """Device table of one grouped weight-gradient launch (sar_gemm_wgrad_group_plan runs on the host; uploaded once).
    probs: tuples (X, dY, dW, db | None, in_dim, out_dim, R, lda, ldb, ldc, splitk).  Raises ValueError when a problem is not
    eligible for the bf16 GEMM path (the caller launches them one by one then)."""

    def __init__(self, probs):
        import torch
        n = len(probs)
        arr = (GemmWgrad * n)()
        for a, q in zip(arr, probs):
            a.X, a.dY, a.dW, a.db = q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), (None if q[3] is None else q[3].data_ptr())
            a.in_dim, a.out_dim, a.R, a.lda, a.ldb, a.ldc, a.splitk = (int(v) for v in q[4:11])
            a.pad_ = int(q[11]) if len(q) > 11 else 0                              # this problem's gradient scale: gscale * 2^-pad_
        buf = ctypes.create_string_buffer(n * 64)
        tb = ctypes.c_int(0)
        rc = lib().fn["sar_gemm_wgrad_group_plan"](arr, n, buf, ctypes.byref(tb))
        if rc != 0:
            raise ValueError("sar_gemm_wgrad_group_plan: status %d (a problem is not on the bf16 GEMM path)" % rc)
        self.keep = [t for q in probs for t in q[:4]]                          # the tensors whose addresses the table holds
        self.n, self.blocks = n, tb.value
        self.table = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).cuda()

    def launch(self, gscale=0.0):
        """gscale > 0: single-pass fp16 operands, dY times gscale (the split-precision mode's backward arithmetic)."""
        call("sar_gemm_wgrad_group_scaled_f32", self.table, self.n, self.blocks, float(gscale))


class WgradConv(ctypes.Structure):
    """include/sarnet.h: sarWgradConv."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("x", "dy", "dw", "dbias")] + \
               [(n, ctypes.c_int) for n in ("B", "H", "W", "Cin", "Ho", "Wo", "Cout", "K", "s", "pt", "pl")] + [("ginv", ctypes.c_float)]


class WgradReduce(ctypes.Structure):
    """include/sarnet.h: sarWgradReduce."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("convs_dev", "tiles_dev", "slices_dev", "slab")] + [(n, ctypes.c_int) for n in ("n_tiles", "K", "tile_ci", "tile_co")]


class WgradPlan:
    """Device tables of one batched weight-gradient launch (sar_wgrad_plan_bf16 runs on the host; uploaded once)."""
    MAX_ITEMS = 16384

    @staticmethod
    def plan_host(descs, target_blocks, tile=(1, 1), max_items=MAX_ITEMS):
        """descs: dicts with the sarWgradConv fields (pointers: tensors, ints or None), all of equal K; tile = (32-channel planes
        per block on the input side, on the output side).  -> (convs table bytes, items table bytes, n_items, launch4); no
        device access."""
        n = len(descs)
        arr = (WgradConv * n)()
        for a, d in zip(arr, descs):
            for k in ("x", "dy", "dw", "dbias"):
                t = d[k]
                setattr(a, k, t.data_ptr() if hasattr(t, "data_ptr") else t)
            for k in ("B", "H", "W", "Cin", "Ho", "Wo", "Cout", "K", "s", "pt", "pl"):
                setattr(a, k, int(d[k]))
            a.ginv = float(d.get("ginv", 1.0))
        cbuf = ctypes.create_string_buffer(n * 128)
        ibuf = ctypes.create_string_buffer(max_items * 32)
        ni, l4 = ctypes.c_int(0), (ctypes.c_int * 8)()
        ntile_max = sum((int(d["Cin"]) // (32 * int(tile[0]))) * (int(d["Cout"]) // (32 * int(tile[1]))) for d in descs)
        tbuf = ctypes.create_string_buffer(max(ntile_max, 1) * 32)
        sbuf = (ctypes.c_int * max_items)()
        rc = lib().fn["sar_wgrad_plan_bf16"](arr, n, int(target_blocks), int(tile[0]), int(tile[1]), cbuf, ibuf, max_items, ctypes.byref(ni), l4, tbuf, sbuf)
        if rc != 0:
            raise RuntimeError("sar_wgrad_plan_bf16 failed with status %d" % rc)
        # + the reduction's tables (include/sarnet.h, "Flush (round 3)"): tile rows, slab index of every slice
        return cbuf.raw, ibuf.raw[:ni.value * 32], ni.value, list(l4), tbuf.raw[:l4[4] * 32], bytes(sbuf)[:l4[5] * 4]

    def __init__(self, descs, target_blocks, tile=(1, 1), sfx="bf16"):
        import torch
        self.entry = "sar_conv2d_wgrad_batch_" + sfx                                # "bf16" | "f16": the 2-byte type of the operands
        self.keep = [d[k] for d in descs for k in ("x", "dy", "dw", "dbias")]      # the tensors whose addresses the tables hold
        self.K = int(descs[0]["K"])
        convs, items, self.n_items, l4, tiles, slices = self.plan_host(descs, target_blocks, tile)
        self.launch4 = (ctypes.c_int * 8)(*l4)
        self.lds, self.np, self.nbuf = l4[0], l4[1], l4[2]
        self.n_tiles, self.n_slabs, self.slab_floats = l4[4], l4[5], l4[6]
        self.convs = torch.frombuffer(bytearray(convs), dtype=torch.uint8).cuda()
        self.items = torch.frombuffer(bytearray(items), dtype=torch.uint8).cuda()
        # partial-tile slabs of the items that share a gradient tile + the reduction's tables
        self.tiles = torch.frombuffer(bytearray(tiles), dtype=torch.uint8).cuda() if self.n_tiles else None
        self.slices = torch.frombuffer(bytearray(slices), dtype=torch.uint8).cuda() if self.n_slabs else None
        self.slab = torch.empty(self.n_slabs * self.slab_floats, dtype=torch.float32, device="cuda") if self.n_slabs else None

    def launch(self, defer_reduce=False):
        """defer_reduce: the partial tiles stay in the slabs; the caller sums them for several plans at once (reduce_multi)."""
        import torch
        self.launch4[7] = 1 if defer_reduce else 0
        rc = lib().fn[self.entry](self.convs.data_ptr(), self.items.data_ptr(), self.n_items, self.K, self.launch4, ptr(self.tiles), ptr(self.slices),
                                  ptr(self.slab), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("%s failed with status %d" % (self.entry, rc))

    @staticmethod
    def reduce_multi(plans, sfx="bf16", overwrite=False):
        """ONE reduction launch for the plans launched with defer_reduce (sar_wgrad_reduce_multi_*; four plans per launch).
        overwrite: dw = the sums instead of dw += (the caller knows dw is zero and has no other writer: the train step)."""
        import torch
        todo = [p for p in plans if p.n_tiles]
        for i in range(0, len(todo), 4):
            grp = todo[i:i + 4]
            arr = (WgradReduce * len(grp))()
            for a, p in zip(arr, grp):
                a.convs_dev, a.tiles_dev, a.slices_dev, a.slab = p.convs.data_ptr(), p.tiles.data_ptr(), p.slices.data_ptr(), p.slab.data_ptr()
                a.n_tiles, a.K, a.tile_ci, a.tile_co = p.n_tiles, p.K + (256 if overwrite else 0), (p.launch4[3] >> 4) & 15, p.launch4[3] & 15
            rc = lib().fn["sar_wgrad_reduce_multi_" + sfx](arr, len(grp), torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise RuntimeError("sar_wgrad_reduce_multi_%s failed with status %d" % (sfx, rc))


def bn_bins():
    return lib().fn["sar_bn_bins"]()


def stem16_wgrad_workspace_bytes(C0):
    return lib().fn["sar_stem16_wgrad_workspace_bytes"](C0)


def bn_workspace_doubles(C):
    """Length (in float64 elements) of the BatchNorm reduction workspace for C channels."""
    return (lib().fn["sar_bn_workspace_bytes"](C) + 7) // 8


def ctc_workspace_bytes(B, L, max_label_len):
    return lib().fn["sar_ctc_workspace_bytes"](B, L, max_label_len)
