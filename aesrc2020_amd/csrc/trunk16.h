// Shared pieces of the bf16 trunk kernels (conv_bf16.hip, stem_bf16.hip): bf16 packing helpers, the BatchNorm-as-a-consumer-sees-it
// plumbing (statistic bins -> constants in every consumer's prologue) and the LDS transpose-read.
#pragma once
#include "common.h"
#ifndef TSTAMP
#define TSTAMP(slot) do {} while (0)    // (conv_bf16.hip defines it in the -DSAR_TIMING build of tools/)
#endif

namespace {

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) short s16x2;

// The 2-byte storage / MFMA operand type of the trunk.  conv_bf16.hip and stem_bf16.hip are compiled three times: as they are (bf16: BASELINE
// configs[1]; entry points sar_*_bf16 / sar_stem16_*), with -DSAR_HALF=1 (IEEE fp16: BASELINE configs[4]; entry points sar_*_f16 /
// sar_stem16_*_f16) and with -DSAR_X3=1 (the fp32-class mode, round 4; entry points sar_*_x3 / sar_stem16_*_x3).  fp16 vs bf16: same bytes,
// same MFMA rate (v_mfma_f32_32x32x16_f16), three more mantissa bits; fp16's narrow exponent is why that build carries a static gradient
// scale (BnP::ginv, WgConv::ginv undo it where parameter gradients leave the trunk).
//
// SAR_X3 -- split precision on the round-3 kernel architecture.  north_star wants logits within 1e-3 of the reference's fp32 graph
// (model.py:197, resnet.py:113-122); single-pass 2-byte operands measure 6e-3 (fp16) / 2e-2 (bf16), and tools/recipe_emulator.py shows that
// no partial recipe (exact weights, fp32 storage, ...) gets below 1.7e-3: BOTH operands of every trunk conv need fp32-class mantissas.  So
// every MFMA operand of the trunk is a PAIR OF PLANES, hi = bf16(x) and lo = bf16(x - hi) (16 mantissa bits), and a product block is three
// MFMAs, lo*hi + hi*lo + hi*hi into one fp32 accumulator (the dropped lo*lo term is 2^-18 relative).  In HBM the trunk's activations and
// gradients are fp32 tensors (r4 v9; until then they were plane pairs as well, the same bytes: every epilogue split, every fill joined,
// transformed and split again); the conv fill splits what it has transformed into the two LDS planes of the patch, the weight packing
// splits the kernels into interleaved hi / lo fragments: no conversion in any k-loop.  Operands that only the weight gradients read (the activated tensors the fills write back, the
// BatchNorm-backward tensors) are kept as ONE fp16 tensor instead (gradient operands times the static scale the fp16 build uses), and the
// weight gradients run the fp16 build's single-pass kernels on them: 2^-12 per product on a sum of ~10^5 products, far inside the 5e-3 the
// parity tests allow a gradient tensor, at half the bytes of planes.
#ifndef SAR_HALF
#define SAR_HALF 0
#endif
#ifndef SAR_X3
#define SAR_X3 0
#endif
#if SAR_HALF && SAR_X3
#error "SAR_HALF and SAR_X3 are different builds"
#endif
#if SAR_HALF
typedef _Float16 h16;
#define MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define SAR_T16(pre) pre##f16
#define SAR_STEM(name) name##_f16
#elif SAR_X3
typedef __bf16 h16;
#define MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define SAR_T16(pre) pre##x3
#define SAR_STEM(name) name##_x3
#else
typedef __bf16 h16;
#define MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define SAR_T16(pre) pre##bf16
#define SAR_STEM(name) name
#endif
constexpr bool X3 = SAR_X3 != 0;
constexpr int NPL = X3 ? 2 : 1;  // planes per 2-byte tensor
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(2))) h16 h16x2;

constexpr int NB = 32;          // statistic bins per BatchNorm (fp32 atomics; contention = blocks / 32 per address; r4 v9: 8 / 16 / 32 = 2.257 / 2.253 / 2.233 ms per step)
constexpr int CMAX = 512;       // max channels of a BatchNorm / conv side

#if SAR_HALF
__device__ __forceinline__ float h_lo(unsigned u) { return (float)__builtin_bit_cast(h16x2, u)[0]; }   // v_cvt_f32_f16 (sdwa for the high half)
__device__ __forceinline__ float h_hi(unsigned u) { return (float)__builtin_bit_cast(h16x2, u)[1]; }
// LDS add of a 2-byte pair (stem weight gradient's scatter): ds_pk_add_f16
__device__ __forceinline__ void lds_pk_add(unsigned char* dst, unsigned val) {
    __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h16x2*)dst, __builtin_bit_cast(h16x2, val));
}
#else
__device__ __forceinline__ float h_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float h_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void lds_pk_add(unsigned char* dst, unsigned val) {                        // ds_pk_add_bf16
    __builtin_amdgcn_ds_atomic_fadd_v2bf16((__attribute__((address_space(3))) s16x2*)dst, __builtin_bit_cast(s16x2, val));
}
#endif
// The 2-byte type of the STEM weight gradient's MFMA operands (scattered gradient image, dy image, im2col image; stem_bf16.hip): the
// build's own type, except in split precision, where that product runs single-pass on fp16 like the other weight gradients.
#if SAR_X3
typedef _Float16 w16;
#define MFMA16W __builtin_amdgcn_mfma_f32_32x32x16_f16
__device__ __forceinline__ void lds_pk_add_w(unsigned char* dst, unsigned val) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_;
    __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) f16x2_*)dst, __builtin_bit_cast(f16x2_, val));
}
#else
typedef h16 w16;
#define MFMA16W MFMA16
__device__ __forceinline__ void lds_pk_add_w(unsigned char* dst, unsigned val) { lds_pk_add(dst, val); }
#endif
typedef __attribute__((ext_vector_type(8))) w16 w16x8;
typedef __attribute__((ext_vector_type(2))) w16 w16x2;
__device__ __forceinline__ float w_lo(unsigned u) { return (float)__builtin_bit_cast(w16x2, u)[0]; }
__device__ __forceinline__ float w_hi(unsigned u) { return (float)__builtin_bit_cast(w16x2, u)[1]; }
__device__ __forceinline__ unsigned wpack2(float a, float b) {
    w16x2 p; p[0] = (w16)a; p[1] = (w16)b;
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    h16x2 p; p[0] = (h16)a; p[1] = (h16)b;                  // v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (RNE)
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void unpack8(const uint4 q, float (&f)[8]) {
    f[0] = h_lo(q.x); f[1] = h_hi(q.x); f[2] = h_lo(q.y); f[3] = h_hi(q.y);
    f[4] = h_lo(q.z); f[5] = h_hi(q.z); f[6] = h_lo(q.w); f[7] = h_hi(q.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
}
// SAR_X3: eight fp32 values <-> their 16-byte pieces of the hi and the lo plane
__device__ __forceinline__ void split8(const float (&f)[8], uint4& hi, uint4& lo) {
    hi = pack8(f);
    float h[8], r[8];
    unpack8(hi, h);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f[j] - h[j];
    lo = pack8(r);
}
// SAR_X3, the HBM side (r4 v9): the trunk's activations and gradients are fp32 TENSORS -- the four bytes per element a (hi, lo) pair of bf16
// planes took.  A producer stores what it has; the consumer's fill splits once, for the LDS planes the MFMAs read (the planes in HBM cost
// a split in every epilogue and a join + split in every fill: 24 of a 16-byte piece's ~150 instructions, 24 per epilogue thread -- and a
// block's life IS its instruction count).  Eight consecutive elements = two 16-byte pieces.
__device__ __forceinline__ void f32x8_of(const uint4 a, const uint4 b, float (&f)[8]) {
    f[0] = __uint_as_float(a.x); f[1] = __uint_as_float(a.y); f[2] = __uint_as_float(a.z); f[3] = __uint_as_float(a.w);
    f[4] = __uint_as_float(b.x); f[5] = __uint_as_float(b.y); f[6] = __uint_as_float(b.z); f[7] = __uint_as_float(b.w);
}
__device__ __forceinline__ uint4 ld16f(const u16* t, size_t e) { return *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(t) + e); }   // four fp32 at element e
__device__ __forceinline__ void st32f(u16* t, size_t e, const float (&f)[8], bool nt = false) {                                                          // eight fp32 at element e
    typedef __attribute__((ext_vector_type(4))) float f4v;
    f4v* d = reinterpret_cast<f4v*>(reinterpret_cast<float*>(t) + e);
    const f4v a = {f[0], f[1], f[2], f[3]}, b = {f[4], f[5], f[6], f[7]};
    if (nt) { __builtin_nontemporal_store(a, d); __builtin_nontemporal_store(b, d + 1); } else { d[0] = a; d[1] = b; }
}
// eight values times `scale` as IEEE fp16 (whatever this build's 2-byte type is): the weight-gradient operands of the SAR_X3 build
__device__ __forceinline__ uint4 pack8_half(const float (&f)[8], float scale) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { f16x2 p; p[0] = (_Float16)(f[2 * j] * scale); p[1] = (_Float16)(f[2 * j + 1] * scale); w[j] = __builtin_bit_cast(unsigned, p); }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// n / d, branch-free: mg = floor(2^32/d) (2^32-1 for d = 1) underestimates the quotient by at most 1
__device__ __forceinline__ unsigned udiv(unsigned n, unsigned mg, unsigned d) {
    const unsigned q = __umulhi(n, mg);
    return q + ((q + 1) * d <= n ? 1u : 0u);
}

// ---------------------------------------------------------------------------------------------------- BatchNorm plumbing
// One BatchNormalization as a consumer kernel sees it (include/sarnet.h: sarBnRef, same layout).
struct BnP {
    const float* gamma; const float* beta;
    const float* bins;          // forward sums [NB][2][C] (sum | sum of squares) of the BN input, or null: use `saved`
    const float* bbins;         // backward sums [NB][2][C] (sum dz | sum dz*xhat)
    float* saved;               // [4][C]: mean, rstd, scale, shift
    float* moving_mean; float* moving_var;
    float* dgamma; float* dbeta;
    float inv_n, unbias, eps, momentum;
    int write;                  // this launch owns the side effects: saved / moving statistics / parameter gradients
    float ginv;                 // 1 / (gradient scale the backward sums in `bbins` carry): applied where dgamma / dbeta are written
    int det;                    // forward sums in `bins` are 64-bit fixed point (units of 2^-28), see fix28()
    int pad_;
};

// Deterministic forward statistics (sarBnRef.det; SAR_BN_DET=1): producers convert every stored value (and its square) to 64-bit fixed
// point and add INTEGERS -- associative, so the sums do not depend on block order, atomic order or how the batch is cut into tiles:
// the same utterances give bit-identical BatchNorm constants in a batch of 32 or duplicated into one of 64.  x * 2^28 is rounded to an
// integer by the add of 1.5 * 2^52 (exact for |x| < 2^23); the bit pattern minus that of the magic number is the integer.
__device__ __forceinline__ long long fix28(float v) {
    const double d = fma((double)v, 268435456.0, 6755399441055744.0);
    return __double_as_longlong(d) - 0x4338000000000000LL;
}
constexpr double FIX28_INV = 1.0 / 268435456.0;

// cst[0..5][cs]: scale, shift, mean, rstd, c2 = S1/N, c3 = S2/N for channels [c0, c0+nc), row stride cs >= nc.  mode 1: forward constants
// (from the bins, or from `saved` when there are none: inference, or a backward consumer); mode 2: backward constants.
__device__ __forceinline__ void bn_prologue(const BnP& bn, int mode, int C, int c0, int nc, float* cst, int cs, bool owner, int nthreads) {
    for (int i = threadIdx.x; i < nc; i += nthreads) {
        const int c = c0 + i;
        float mean, rstd, scale, shift;
        if (mode == 1 && bn.bins != nullptr) {
            double s1 = 0.0, s2 = 0.0;
            if (bn.det) {                                            // kernel-uniform
                const long long* qb = reinterpret_cast<const long long*>(bn.bins);
                long long q1 = 0, q2 = 0;
#pragma unroll
                for (int b = 0; b < NB; ++b) { q1 += qb[(b * 2) * C + c]; q2 += qb[(b * 2 + 1) * C + c]; }
                s1 = (double)q1 * FIX28_INV; s2 = (double)q2 * FIX28_INV;
            } else {
#pragma unroll
                for (int b = 0; b < NB; ++b) { s1 += (double)bn.bins[(b * 2) * C + c]; s2 += (double)bn.bins[(b * 2 + 1) * C + c]; }
            }
            const double mu = s1 * (double)bn.inv_n;
            double var = s2 * (double)bn.inv_n - mu * mu;
            var = var > 0.0 ? var : 0.0;
            mean = (float)mu; rstd = 1.0f / sqrtf((float)var + bn.eps);
            scale = bn.gamma[c] * rstd; shift = bn.beta[c] - mean * scale;
            if (owner && bn.write) {
                bn.saved[c] = mean; bn.saved[C + c] = rstd; bn.saved[2 * C + c] = scale; bn.saved[3 * C + c] = shift;
                if (bn.moving_mean != nullptr) {
                    bn.moving_mean[c] = bn.moving_mean[c] * bn.momentum + mean * (1.f - bn.momentum);
                    bn.moving_var[c] = bn.moving_var[c] * bn.momentum + (float)var * bn.unbias * (1.f - bn.momentum);
                }
            }
        } else {
            mean = bn.saved[c]; rstd = bn.saved[C + c]; scale = bn.saved[2 * C + c]; shift = bn.saved[3 * C + c];
        }
        cst[i] = scale; cst[cs + i] = shift; cst[2 * cs + i] = mean; cst[3 * cs + i] = rstd;
        if (mode == 2) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) { s1 += (double)bn.bbins[(b * 2) * C + c]; s2 += (double)bn.bbins[(b * 2 + 1) * C + c]; }
            cst[4 * cs + i] = (float)(s1 * (double)bn.inv_n); cst[5 * cs + i] = (float)(s2 * (double)bn.inv_n);
            if (owner && bn.write && bn.dgamma != nullptr) { bn.dgamma[c] += (float)s2 * bn.ginv; bn.dbeta[c] += (float)s1 * bn.ginv; }
        }
    }
}

// The same constants, computed by the WHOLE block (round 3; conv_bf16_kernel).  bn_prologue above has thread c walk all 16 bins of
// channel c alone: 32 loads (64 in mode 2) and as many fp64 adds on C threads while the rest of the block waits at the barrier behind it --
// 2 800 cycles of a stage-2 block's 20 400, 8 700 of a stage-0 block's 27 000 (32 channels: half a wave works, 500 blocks wait on the
// same 64 cache lines).  Here G = min(NTHR / C, 4) threads share a channel: thread (c, q) sums the bins q, q + G, ... (NB / G loads per
// sum, one round trip, coalesced), the partial sums meet in LDS scratch `scr` (G x 4 x C x 8 bytes: the patch region, not yet filled), and
// thread c finishes the channel.  Summation order is fixed (bins of a group ascending, groups ascending), the integer path exact.
template <int G>
__device__ __forceinline__ void bn_partials(const BnP& bn, int mode, bool fwd_bins, int C, int c, int q, double* scr) {
    if (fwd_bins) {
        if (bn.det) {                                                // kernel-uniform
            const long long* qb = reinterpret_cast<const long long*>(bn.bins);
            long long q1 = 0, q2 = 0;
#pragma unroll
            for (int b = 0; b < NB / G; ++b) { q1 += qb[((q + b * G) * 2) * C + c]; q2 += qb[((q + b * G) * 2 + 1) * C + c]; }
            reinterpret_cast<long long*>(scr)[(q * 2) * C + c] = q1; reinterpret_cast<long long*>(scr)[(q * 2 + 1) * C + c] = q2;
        } else {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int b = 0; b < NB / G; ++b) { s1 += (double)bn.bins[((q + b * G) * 2) * C + c]; s2 += (double)bn.bins[((q + b * G) * 2 + 1) * C + c]; }
            scr[(q * 2) * C + c] = s1; scr[(q * 2 + 1) * C + c] = s2;
        }
    }
    if (mode == 2) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int b = 0; b < NB / G; ++b) { s1 += (double)bn.bbins[((q + b * G) * 2) * C + c]; s2 += (double)bn.bbins[((q + b * G) * 2 + 1) * C + c]; }
        scr[(2 * G + q * 2) * C + c] = s1; scr[(2 * G + q * 2 + 1) * C + c] = s2;
    }
}
__device__ __forceinline__ void bn_finalize(const BnP& bn, int mode, bool fwd_bins, int G, int C, float* cst, int cs, bool owner, const double* scr) {
    const int tid = threadIdx.x;
    if (tid < C) {
        const int c = tid;
        float mean, rstd, scale, shift;
        if (fwd_bins) {
            double s1 = 0.0, s2 = 0.0;
            if (bn.det) {
                long long q1 = 0, q2 = 0;
                for (int q = 0; q < G; ++q) { q1 += reinterpret_cast<const long long*>(scr)[(q * 2) * C + c]; q2 += reinterpret_cast<const long long*>(scr)[(q * 2 + 1) * C + c]; }
                s1 = (double)q1 * FIX28_INV; s2 = (double)q2 * FIX28_INV;
            } else {
                for (int q = 0; q < G; ++q) { s1 += scr[(q * 2) * C + c]; s2 += scr[(q * 2 + 1) * C + c]; }
            }
            const double mu = s1 * (double)bn.inv_n;
            double var = s2 * (double)bn.inv_n - mu * mu;
            var = var > 0.0 ? var : 0.0;
            mean = (float)mu; rstd = 1.0f / sqrtf((float)var + bn.eps);
            scale = bn.gamma[c] * rstd; shift = bn.beta[c] - mean * scale;
            if (owner && bn.write) {
                bn.saved[c] = mean; bn.saved[C + c] = rstd; bn.saved[2 * C + c] = scale; bn.saved[3 * C + c] = shift;
                if (bn.moving_mean != nullptr) {
                    bn.moving_mean[c] = bn.moving_mean[c] * bn.momentum + mean * (1.f - bn.momentum);
                    bn.moving_var[c] = bn.moving_var[c] * bn.momentum + (float)var * bn.unbias * (1.f - bn.momentum);
                }
            }
        } else {
            mean = bn.saved[c]; rstd = bn.saved[C + c]; scale = bn.saved[2 * C + c]; shift = bn.saved[3 * C + c];
        }
        cst[c] = scale; cst[cs + c] = shift; cst[2 * cs + c] = mean; cst[3 * cs + c] = rstd;
        if (mode == 2) {
            double s1 = 0.0, s2 = 0.0;
            for (int q = 0; q < G; ++q) { s1 += scr[(2 * G + q * 2) * C + c]; s2 += scr[(2 * G + q * 2 + 1) * C + c]; }
            cst[4 * cs + c] = (float)(s1 * (double)bn.inv_n); cst[5 * cs + c] = (float)(s2 * (double)bn.inv_n);
            if (owner && bn.write && bn.dgamma != nullptr) { bn.dgamma[c] += (float)s2 * bn.ginv; bn.dbeta[c] += (float)s1 * bn.ginv; }
        }
    }}

template <int NTHR>
__device__ __forceinline__ void bn_prologue_coop(const BnP& bn, int mode, int C, float* cst, int cs, bool owner, double* scr) {
    const int tid = threadIdx.x;
    const bool fwd_bins = mode == 1 && bn.bins != nullptr;
    const int G = NTHR / C >= 4 ? 4 : NTHR / C >= 2 ? 2 : 1;         // C is a power of two <= NTHR (the launcher checks)
    if (fwd_bins || mode == 2) {
        const int c = tid & (C - 1), q = tid / C;
        if (q < G) {
            if (G == 4) bn_partials<4>(bn, mode, fwd_bins, C, c, q, scr);
            else if (G == 2) bn_partials<2>(bn, mode, fwd_bins, C, c, q, scr);
            else bn_partials<1>(bn, mode, fwd_bins, C, c, q, scr);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    bn_finalize(bn, mode, fwd_bins, G, C, cst, cs, owner, scr);
}

// bn_prologue_coop in two halves, so that the round trip of the bins (written by the producer's atomics: never in this XCD's L2, ~2 800
// cycles) runs underneath the row table, the fill's address arithmetic and the round trip of the fill's own first loads: bn_issue requests
// this thread's share of the bins into registers at kernel entry, bn_finish (called by the fill behind the issue of its first pieces) sums,
// meets the other groups in LDS and writes the constants.  G = 1 (a channel per thread and no spare threads) loads late, in bn_finish.
// (floats: sums 1 in a then b, sums 2 in c then d -- up to 16 bins per thread; 64-bit integers: sums 1 as pairs in a then b, sums 2 in c then d
// -- up to 8 bins per thread, more are read late, in bn_sum_g.  Forward and backward bins are never both read by one launch.)
typedef __attribute__((ext_vector_type(8))) unsigned u32x8;
struct BnRaw { u32x8 a, b, c, d; };   // (vector VALUES, not arrays: nothing an alloca could be made of.)  v[0..7] = a, v[8..15] = b, v[16..23] = c, v[24..31] = d
// (ONE float path for the forward and the backward bins -- the base pointer is selected, not the code: with two look-alike branches the
// optimiser sank their stores into one with a selected ARRAY INDEX, and `r` went to scratch)
template <int G>
__device__ __forceinline__ void bn_issue_g(const BnP& bn, int mode, bool fwd_bins, int C, int c, int q, BnRaw& r) {
    static_assert(NB / G <= 16, "BnRaw holds sixteen bins per thread");
    if (fwd_bins && bn.det) {
        if constexpr (NB / G <= 8) {
            const long long* qp = reinterpret_cast<const long long*>(bn.bins);
#pragma unroll
            for (int b = 0; b < NB / G; ++b) {
                const uint2 t1 = *reinterpret_cast<const uint2*>(qp + ((q + b * G) * 2) * C + c), t2 = *reinterpret_cast<const uint2*>(qp + ((q + b * G) * 2 + 1) * C + c);
                if (b < 4) { r.a[2 * b] = t1.x; r.a[2 * b + 1] = t1.y; r.c[2 * b] = t2.x; r.c[2 * b + 1] = t2.y; }
                else { r.b[2 * b - 8] = t1.x; r.b[2 * b - 7] = t1.y; r.d[2 * b - 8] = t2.x; r.d[2 * b - 7] = t2.y; }
            }
        }
    } else {
        const float* src = mode == 2 ? bn.bbins : bn.bins;
#pragma unroll
        for (int b = 0; b < NB / G; ++b) {
            const unsigned v1 = __float_as_uint(src[((q + b * G) * 2) * C + c]), v2 = __float_as_uint(src[((q + b * G) * 2 + 1) * C + c]);
            if (b < 8) { r.a[b & 7] = v1; r.c[b & 7] = v2; } else { r.b[b & 7] = v1; r.d[b & 7] = v2; }
        }
    }
}
template <int G>
__device__ __forceinline__ void bn_sum_g(const BnP& bn, int mode, bool fwd_bins, int C, int c, int q, const BnRaw& r, double* scr) {
    if (fwd_bins && bn.det) {
        long long q1 = 0, q2 = 0;
        if constexpr (NB / G <= 8) {
#pragma unroll
            for (int b = 0; b < NB / G; ++b) {
                if (b < 4) { q1 += (long long)(((unsigned long long)r.a[2 * b + 1] << 32) | r.a[2 * b]); q2 += (long long)(((unsigned long long)r.c[2 * b + 1] << 32) | r.c[2 * b]); }
                else { q1 += (long long)(((unsigned long long)r.b[2 * b - 7] << 32) | r.b[2 * b - 8]); q2 += (long long)(((unsigned long long)r.d[2 * b - 7] << 32) | r.d[2 * b - 8]); }
            }
        } else {                                                    // (more 64-bit bins than BnRaw holds: read here)
            const long long* qb = reinterpret_cast<const long long*>(bn.bins);
#pragma unroll
            for (int b = 0; b < NB / G; ++b) { q1 += qb[((q + b * G) * 2) * C + c]; q2 += qb[((q + b * G) * 2 + 1) * C + c]; }
        }
        reinterpret_cast<long long*>(scr)[(q * 2) * C + c] = q1; reinterpret_cast<long long*>(scr)[(q * 2 + 1) * C + c] = q2;
    } else {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int b = 0; b < NB / G; ++b) {
            s1 += (double)__uint_as_float(b < 8 ? r.a[b & 7] : r.b[b & 7]); s2 += (double)__uint_as_float(b < 8 ? r.c[b & 7] : r.d[b & 7]);
        }
        const int o = mode == 2 ? 2 * G : 0;
        scr[(o + q * 2) * C + c] = s1; scr[(o + q * 2 + 1) * C + c] = s2;
    }
}
template <int NTHR>
__device__ __forceinline__ int bn_groups(int C) { return NTHR / C >= 4 ? 4 : NTHR / C >= 2 ? 2 : 1; }
// (C is a compile-time constant of the calling kernel: with G chosen at run time, parts of `r` ended up in scratch -- and a wait for the
// loaded value in front of the scratch store, i.e. no early issue at all)
template <int NTHR, int C>
__device__ __forceinline__ void bn_issue(const BnP& bn, int mode, BnRaw& r) {
    constexpr int G = NTHR / C >= 4 ? 4 : NTHR / C >= 2 ? 2 : 1;
    const bool fwd_bins = mode == 1 && bn.bins != nullptr;
    const int tid = threadIdx.x, c = tid & (C - 1), q = tid / C;
    if constexpr (G > 1) {
        if ((fwd_bins || mode == 2) && q < G) bn_issue_g<G>(bn, mode, fwd_bins, C, c, q, r);
    }
}

// (the three fields of the BatchNorm it reads, handed over one by one: the planned conv kernels keep copies of them in the first piece of their argument block)
template <int NTHR, int C>
__device__ __forceinline__ void bn_issue(const float* bins, const float* bbins, int det, int mode, BnRaw& r) {
    BnP bn;
    bn.bins = bins; bn.bbins = bbins; bn.det = det;
    bn_issue<NTHR, C>(bn, mode, r);
}
template <int NTHR, int C>
__device__ __forceinline__ void bn_finish(const BnP& bn, int mode, float* cst, int cs, bool owner, const BnRaw& r, double* scr) {
    constexpr int G = NTHR / C >= 4 ? 4 : NTHR / C >= 2 ? 2 : 1;
    const int tid = threadIdx.x;
    const bool fwd_bins = mode == 1 && bn.bins != nullptr;
    if (fwd_bins || mode == 2) {
        const int c = tid & (C - 1), q = tid / C;
        if (q < G) {
            if constexpr (G > 1) bn_sum_g<G>(bn, mode, fwd_bins, C, c, q, r, scr);
            else bn_partials<1>(bn, mode, fwd_bins, C, c, q, scr);
        }
        TSTAMP(19);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        TSTAMP(20);
    }
    bn_finalize(bn, mode, fwd_bins, G, C, cst, cs, owner, scr);
    TSTAMP(21);
}

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ s16x4 lds_tr(const unsigned char* base, unsigned off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
}

}  // namespace
