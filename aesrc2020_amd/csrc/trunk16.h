// Shared pieces of the bf16 trunk kernels (conv_bf16.hip, stem_bf16.hip): bf16 packing helpers, the BatchNorm-as-a-consumer-sees-it
// plumbing (statistic bins -> constants in every consumer's prologue) and the LDS transpose-read.
#pragma once
#include "common.h"

namespace {

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) short s16x2;

constexpr int NB = 16;          // statistic bins per BatchNorm (fp32 atomics; contention = blocks / 16 per address)
constexpr int CMAX = 512;       // max channels of a BatchNorm / conv side

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 p; p[0] = (__bf16)a; p[1] = (__bf16)b;           // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void unpack8(const uint4 q, float (&f)[8]) {
    f[0] = bf_lo(q.x); f[1] = bf_hi(q.x); f[2] = bf_lo(q.y); f[3] = bf_hi(q.y);
    f[4] = bf_lo(q.z); f[5] = bf_hi(q.z); f[6] = bf_lo(q.w); f[7] = bf_hi(q.w);
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
    int pad_;
};

// cst[0..5][cs]: scale, shift, mean, rstd, c2 = S1/N, c3 = S2/N for channels [c0, c0+nc), row stride cs >= nc.  mode 1: forward constants
// (from the bins, or from `saved` when there are none: inference, or a backward consumer); mode 2: backward constants.
__device__ __forceinline__ void bn_prologue(const BnP& bn, int mode, int C, int c0, int nc, float* cst, int cs, bool owner, int nthreads) {
    for (int i = threadIdx.x; i < nc; i += nthreads) {
        const int c = c0 + i;
        float mean, rstd, scale, shift;
        if (mode == 1 && bn.bins != nullptr) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) { s1 += (double)bn.bins[(b * 2) * C + c]; s2 += (double)bn.bins[(b * 2 + 1) * C + c]; }
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
            if (owner && bn.write && bn.dgamma != nullptr) { bn.dgamma[c] += (float)s2; bn.dbeta[c] += (float)s1; }
        }
    }
}

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ s16x4 lds_tr(const unsigned char* base, unsigned off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
}

}  // namespace
