"""Arithmetic modes the GPU tests select (aesrc2020_amd._lib): the autouse fixture of conftest.py starts every test from round 1's
fp32-storage trunk; `mode(name)` switches for the body of a `with` and restores that baseline.

  "fp32"       fp32-storage trunk (csrc/conv_igemm.hip), exact fp32-MFMA convs, fp32 GEMMs
  "bf16x3_r1"  the same kernels, convs in split precision (round 1's fp32-class mode)
  "x3"         THE PACKAGE DEFAULT (_lib.DEFAULT_MODE): split-precision trunk of csrc/conv_bf16.hip (-DSAR_X3) + split-precision GEMMs + fp16
               MFMA recurrence at hidden_dim 256 -- what a SAR_Net(...) built by the reference's train.py runs in
  "fp16" / "bf16"  the single-pass 2-byte trunks with bf16 GEMMs (BASELINE configs[4] / configs[1]'s names)
"""
import contextlib


@contextlib.contextmanager
def mode(name):
    from aesrc2020_amd import _lib
    old_conv = _lib.get_conv_math()
    trunk, gemm, conv = {"fp32": ("fp32", "fp32", "fp32"), "bf16x3_r1": ("fp32", "fp32", "bf16x3"), "x3": ("bf16x3", "bf16x3", None),
                         "fp16": ("fp16", "bf16", None), "bf16": ("bf16", "bf16", None)}[name]
    _lib.set_trunk_dtype(trunk); _lib.set_gemm_math(gemm)
    if conv is not None:
        _lib.set_conv_math(conv)
    try:
        yield name
    finally:
        _lib.set_trunk_dtype("fp32"); _lib.set_gemm_math("fp32"); _lib.set_conv_math(old_conv)


def runs_shipped_kernels(model):
    """True when the model's trunk really is on the split-precision kernels (res_filters 32 / 64); other widths fall back to the fp32-storage ones."""
    return bool(getattr(model.trunk, "x3", False))
