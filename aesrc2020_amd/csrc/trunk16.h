// Shared pieces of the bf16 trunk kernels (conv_bf16.hip, stem_bf16.hip): bf16 packing helpers, the BatchNorm-as-a-consumer-sees-it
// plumbing (statistic bins -> constants in every consumer's prologue) and the LDS transpose-read.
#pragma once
#include "common.h"

namespace {

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) short s16x2;

// The 2-byte storage / MFMA operand type of the trunk.  conv_bf16.hip and stem_bf16.hip are compiled twice: as they are (bf16: BASELINE
// configs[1]; entry points sar_*_bf16 / sar_stem16_*) and with -DSAR_HALF=1 (IEEE fp16: BASELINE configs[4]; entry points sar_*_f16 /
// sar_stem16_*_f16).  Same bytes, same MFMA rate (v_mfma_f32_32x32x16_f16), three more mantissa bits; fp16's narrow exponent is why
// that build carries a static gradient scale (BnP::ginv, WgConv::ginv undo it where parameter gradients leave the trunk).
#ifndef SAR_HALF
#define SAR_HALF 0
#endif
#if SAR_HALF
typedef _Float16 h16;
#define MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define SAR_T16(pre) pre##f16
#define SAR_STEM(name) name##_f16
#else
typedef __bf16 h16;
#define MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define SAR_T16(pre) pre##bf16
#define SAR_STEM(name) name
#endif
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(2))) h16 h16x2;

constexpr int NB = 16;          // statistic bins per BatchNorm (fp32 atomics; contention = blocks / 16 per address)
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

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ s16x4 lds_tr(const unsigned char* base, unsigned off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
}

}  // namespace
