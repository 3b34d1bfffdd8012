// Generic fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain),
// used by Dense fwd/bwd, the GRU input/recurrent-weight GEMMs and the VLAD assignment conv.
//   C[M,N] (+)= act( op(A)[M,K] * op(B)[K,N] + bias[N] )
// Layout flags: a_kc = A stored (M,K) row-major (k contiguous); !a_kc = A stored (K,M).
//               b_kc = B stored (N,K) row-major (k contiguous); !b_kc = B stored (K,N).
// Tile 64x64x16, 4 waves (2x2), one 32x32 accumulator per wave. LDS tiles are k-major
// ([k][m] / [k][n]) so the MFMA operand reads (lane&31 -> m/n, lane>>5 -> k) are conflict-free.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDP = 68;  // LDP: padded row (floats), keeps 16-B alignment

// tanh on v_exp_f32 + v_rcp_f32 (absolute error ~1e-7; libm's tanhf is ~100 instructions per value, 16 values per lane in the epilogue of a
// launch whose whole block lives ~6 us: the tanh Dense layers cost 8.1 us against 6.8 for the same GEMM without it)
// (the one-line form 1 - 2 / (1 + e^2a) cancels near zero -- 1e-3 relative at |a| = 1e-4 -- and is not odd.  Here: the odd Taylor
// polynomial below |a| = 1/16 (truncation 3e-9 relative), (1 - t) / (1 + t) with t = e^-2|a| <= 0.8825 above it (5e-7 relative), sign copied.)
__device__ __forceinline__ float fast_tanh(float a) {
    const float x = fabsf(a), x2 = a * a;
    const float small = a * fmaf(x2, fmaf(x2, 0.13333334f, -0.33333334f), 1.f);
    const float t = __builtin_amdgcn_exp2f(-2.88539008f * x);
    const float big = copysignf((1.f - t) * __builtin_amdgcn_rcpf(1.f + t), a);
    return x < 0.0625f ? small : big;
}

template <bool KC>
__device__ __forceinline__ void load_tile(float (&r)[4], const float* __restrict__ P, int ld, int mn0, int k0,
                                          int MN, int K, int tid, bool vec_ok) {
    // Tile is 64 (m or n) x 16 (k); each of 256 threads fetches 4 consecutive elements along the contiguous dim.
    if (KC) {
        const int mn = mn0 + (tid >> 2), k = k0 + (tid & 3) * 4;
        if (mn < MN && vec_ok && k + 3 < K) {
            const float4 v = *reinterpret_cast<const float4*>(P + (long)mn * ld + k);
            r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = (mn < MN && k + i < K) ? P[(long)mn * ld + k + i] : 0.f;
        }
    } else {
        const int k = k0 + (tid >> 4), mn = mn0 + (tid & 15) * 4;
        if (k < K && vec_ok && mn + 3 < MN) {
            const float4 v = *reinterpret_cast<const float4*>(P + (long)k * ld + mn);
            r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = (k < K && mn + i < MN) ? P[(long)k * ld + mn + i] : 0.f;
        }
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile(const float (&r)[4], float (*S)[LDP], int tid) {
    if (KC) {
        const int mn = tid >> 2, k = (tid & 3) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) S[k + i][mn] = r[i];
    } else {
        const int k = tid >> 4, mn = (tid & 15) * 4;
        *reinterpret_cast<float4*>(&S[k][mn]) = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// Up to four problems of equal shape in one launch (grid.z = batch x split-K slices): the two directions of a BiGRU's input
// projections, the four weight gradients of its kernels -- launches of ~1 us of matrix time each that otherwise queue behind one another.
struct GemmBatch { const float* A[4]; const float* B[4]; float* C[4]; const float* bias[4]; float* colsum[4]; };

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmBatch gb, int M,
                                                       int N, int K, int lda, int ldb, int ldc, int act,
                                                       int accumulate, int splitk) {
    const int bz = blockIdx.z / splitk, kz = blockIdx.z - bz * splitk;
    const float* __restrict__ A = gb.A[bz]; const float* __restrict__ B = gb.B[bz];
    float* __restrict__ C = gb.C[bz]; const float* __restrict__ bias = gb.bias[bz]; float* __restrict__ colsum = gb.colsum[bz];
    __shared__ __attribute__((aligned(16))) float As2[2][BK][LDP];
    __shared__ __attribute__((aligned(16))) float Bs2[2][BK][LDP];
    float (*As)[LDP] = As2[0];
    float (*Bs)[LDP] = Bs2[0];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    // split-K: slice z handles k-tiles [kt0, kt1)
    const int nkt = (K + BK - 1) / BK;
    const int per = (nkt + splitk - 1) / splitk;
    const int kt0 = kz * per, kt1 = min(nkt, kt0 + per);
    const int bcol = n0 + wn * 32 + (lane & 31);                      // bias requested up front (see gemm_bf16_tile)
    const float bv = (bias != nullptr && kz == 0 && bcol < N) ? bias[bcol] : 0.f;
    const bool a_vec = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool b_vec = ((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float ra[4], rb[4];
    // colsum (weight-gradient GEMMs, B stored (K,N)): the first row of blocks also reduces its B tiles over k -- the bias
    // gradient sum_rows dY comes for free with dW = X^T dY (B is dY) instead of a separate pass over dY
    const bool do_colsum = !B_KC && colsum != nullptr && blockIdx.y == 0;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    // Interior blocks of aligned problems (every Dense / GRU-projection GEMM of the model): branch-free float4 loads, two
    // register sets (a tile is loaded two tiles before its MFMAs), two LDS buffers and ONE barrier per k-tile, the staging of
    // tile t+1 in the same basic block as the MFMAs of tile t; k-tiles past the end are staged as zeros so that the two-half
    // loop has no conditional half.  (The generic loop below issues its loads 8 MFMAs = 512 cycles ahead of their use and
    // waits on them with vmcnt(0) every tile.)
    const bool fast = a_vec && b_vec && m0 + BM <= M && n0 + BN <= N && (K % BK) == 0 && kt0 < kt1;   // block-uniform
    if (fast) {
        const int last = kt1 - 1;
        float4 fa[2], fb[2];
        auto fload = [&](int set, int kt) {
            const int k0 = min(kt, last) * BK;
            if (A_KC) fa[set] = *reinterpret_cast<const float4*>(A + (long)(m0 + (tid >> 2)) * lda + k0 + (tid & 3) * 4);
            else      fa[set] = *reinterpret_cast<const float4*>(A + (long)(k0 + (tid >> 4)) * lda + m0 + (tid & 15) * 4);
            if (B_KC) fb[set] = *reinterpret_cast<const float4*>(B + (long)(n0 + (tid >> 2)) * ldb + k0 + (tid & 3) * 4);
            else      fb[set] = *reinterpret_cast<const float4*>(B + (long)(k0 + (tid >> 4)) * ldb + n0 + (tid & 15) * 4);
        };
        auto fstore = [&](int set, int buf, int kt) {
            const float live = kt <= last ? 1.f : 0.f;
            const float4 a = fa[set], b = fb[set];
            const float va[4] = {a.x * live, a.y * live, a.z * live, a.w * live}, vb[4] = {b.x * live, b.y * live, b.z * live, b.w * live};
            store_tile<A_KC>(va, As2[buf], tid);
            store_tile<B_KC>(vb, Bs2[buf], tid);
            if (do_colsum) { cs[0] += vb[0]; cs[1] += vb[1]; cs[2] += vb[2]; cs[3] += vb[3]; }
        };
        auto fmma = [&](int buf) {
#pragma unroll
            for (int k = 0; k < BK; k += 2) {
                const float a = As2[buf][k + (lane >> 5)][wm * 32 + (lane & 31)];
                const float b = Bs2[buf][k + (lane >> 5)][wn * 32 + (lane & 31)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        };
        fload(0, kt0); fload(1, kt0 + 1);
        fstore(0, 0, kt0); fload(0, kt0 + 2);
        __syncthreads();
        for (int kt = kt0; kt < kt1; kt += 2) {
            fmma(0); fstore(1, 1, kt + 1); fload(1, kt + 3);
            __syncthreads();
            fmma(1); fstore(0, 0, kt + 2); fload(0, kt + 4);
            __syncthreads();
        }
    } else if (kt0 < kt1) {
        load_tile<A_KC>(ra, A, lda, m0, kt0 * BK, M, K, tid, a_vec);
        load_tile<B_KC>(rb, B, ldb, n0, kt0 * BK, N, K, tid, b_vec);
    }
    for (int kt = fast ? kt1 : kt0; kt < kt1; ++kt) {
        store_tile<A_KC>(ra, As, tid);
        store_tile<B_KC>(rb, Bs, tid);
        if (do_colsum) { cs[0] += rb[0]; cs[1] += rb[1]; cs[2] += rb[2]; cs[3] += rb[3]; }
        __syncthreads();
        if (kt + 1 < kt1) {
            load_tile<A_KC>(ra, A, lda, m0, (kt + 1) * BK, M, K, tid, a_vec);
            load_tile<B_KC>(rb, B, ldb, n0, (kt + 1) * BK, N, K, tid, b_vec);
        }
#pragma unroll
        for (int k = 0; k < BK; k += 2) {
            const float a = As[k + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[k + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if (do_colsum) {                                  // block-uniform; Bs is free after the loop's last barrier
        *reinterpret_cast<float4*>(&Bs[tid >> 4][(tid & 15) * 4]) = make_float4(cs[0], cs[1], cs[2], cs[3]);
        __syncthreads();
        if (tid < BN && n0 + tid < N) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < BK; ++k) t += Bs[k][tid];
            atomicAdd(colsum + n0 + tid, t);
        }
    }
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= M) continue;
        float v = acc[r] + bv;
        float* dst = C + (long)row * ldc + col;
        if (splitk > 1) {
            atomicAdd(dst, v);
        } else {
            if (act == ACT_TANH) v = fast_tanh(v);
            else if (act == ACT_RELU) v = fmaxf(v, 0.f);
            if (accumulate) v += *dst;
            *dst = v;
        }
    }
}


// ---- bf16 MFMA path (sar_set_gemm_math(1); the "bf16" bench mode) ----------------------------------------------------------------
// Same contract as gemm_f32_kernel for problems whose M, N are multiples of 64 and K of 64, with 16-byte aligned rows: the fp32
// operands are rounded to bf16 while they are staged into LDS (one v_cvt_pk_bf16_f32 per two elements, RNE) and multiplied on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- 1/16 of the fp32-MFMA issue time; bias, activation, accumulate, split-K
// and the fused column sum stay fp32.  Tile 64 x 64 x 64, four waves (2 x 2), two LDS buffers, one barrier per k-tile.
// A k-contiguous operand is staged [mn][k] (144-byte rows) and fetched with ds_read_b128; an operand stored (K, MN) is staged
// as it lies, [k][mn], and fetched through ds_read_b64_tr_b16 (hardware transpose) -- no element-wise scatter for any layout.
typedef __attribute__((ext_vector_type(8))) __bf16 gbf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 gbf16x2;
typedef __attribute__((ext_vector_type(4))) short gs16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 gf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 gf16x8;
constexpr int GB_K = 64, GB_RS = 144;          // k-tile; LDS row stride in bytes (64 bf16 + 16: conflict-free b128 column reads)

__device__ __forceinline__ unsigned gpack2(float a, float b) {
    gbf16x2 p; p[0] = (__bf16)a; p[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ unsigned gpack2h(float a, float b) {
    gf16x2 p; p[0] = (_Float16)a; p[1] = (_Float16)b;          // RNE
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ gs16x4 g_lds_tr(const unsigned char* base, unsigned off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(base + off));
}

// TAIL: M, N and / or K need not be multiples of 64 (multiples of 4: rows stay 16-byte pieces) -- the bpe-1k CTC output layer (N = 1000
// forward and in its weight gradient, K = 1000 in its input gradient), cfg5's 6 048 rows: pieces past the edge are loaded from a
// clamped address and zeroed, stores past M / N are dropped.  (Element-by-element staging for extents that are not multiples of 4 --
// the 34 / 66 / 10 columns of the VLAD assignment layer -- was built and measured: 24.6 us against the fp32 kernel's 19.6 forward,
// and its weight gradient cost the grouped launch 17 us where it had cost 11.7 alone; those problems stay on the fp32 kernel.)
// One 64 x 64 output tile (tile column bx, tile row by, split-K slice kz); the LDS comes from the calling kernel, so that a kernel
// which holds several instantiations (the grouped weight-gradient launch below) still has ONE set of buffers.
// X3 (sar_set_gemm_math(2), the fp32-class "bf16x3" mode): the fp32 operands are split into hi = bf16(x) and lo = bf16(x - hi) while they are
// staged (two LDS planes per operand) and a product block is three MFMAs, lo*hi + hi*lo + hi*hi -- 2^-17 relative per operand instead of
// 2^-9.  Single-pass 2-byte operands in these GEMMs alone cost 1e-3 (fp16) / 7e-3 (bf16) on the accent probabilities
// (tools/recipe_emulator.py: gemm_x / gemm_w), the whole of north_star's tolerance.
template <bool X3> struct GemmLdsT { unsigned char As[2][X3 ? 2 : 1][64 * GB_RS]; unsigned char Bs[2][X3 ? 2 : 1][64 * GB_RS]; float csum[4][64]; };
typedef GemmLdsT<false> GemmLds;
// (Negative, round 3: 512-thread blocks -- two groups of four waves on the even / odd k-tiles, meeting in LDS -- 7.39 -> 7.15 us for the
// 156-tile launch but 9.9 -> 12.9 for the 468-tile one and the step 1.981 -> 1.996 ms: the launches with more than 256 tiles already had two
// blocks per CU.)
// F16 (round 5: the BACKWARD products of the per-frame Dense / GRU-projection layers in the split-precision mode -- input gradients and the
// grouped weight gradient): ONE pass on IEEE fp16 copies of the operands, each times a power of two while it is staged (x.sa / x.sb: the
// gradient operand carries the trunk's static gradient scale, csrc/conv_bf16.hip), the accumulators times 1 / (sa sb) on the way out.  The
// parameter gradients' precision is set by single-pass fp16 operands since the mixed backward of the trunk (tools/grad_recipe_emulator.py:
// flat gradient 9.6e-4 -> 1.09e-3 with these products too; bf16 operands here: 4.6e-3), the outputs' 1e-3 only binds the forward pass.
// GemmX.A2 / B2 (any mode, whole tiles): a SECOND operand pair along k -- C = A B + A2 B2 in one launch, K = both segments, seg = k-tiles of
// the first (the BiGRU's input gradient: two directions' products into one tensor were two launches, the second a read-modify-write of C).
struct GemmX { float sa, sb; const float* A2; const float* B2; int seg; };
constexpr GemmX GX0{1.f, 1.f, nullptr, nullptr, 0};
template <bool A_KC, bool B_KC, bool TAIL, bool X3 = false, bool F16 = false>
__device__ __forceinline__ void gemm_bf16_tile(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                               const float* __restrict__ bias, float* __restrict__ colsum, int M, int N, int K, int lda, int ldb,
                                               int ldc, int act, int accumulate, int splitk, int bx, int by, int kz, GemmLdsT<X3>* Ls,
                                               const GemmX x = GX0) {
    static_assert(!(X3 && F16), "one arithmetic per instantiation");
    constexpr int NPL = X3 ? 2 : 1;
    GemmLdsT<X3>& L = *Ls;
    unsigned char (&As)[2][NPL][64 * GB_RS] = L.As; unsigned char (&Bs)[2][NPL][64 * GB_RS] = L.Bs; float (&csum)[4][64] = L.csum;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = by * BM, n0 = bx * BN;
    const int nkt = (K + GB_K - 1) / GB_K, per = (nkt + splitk - 1) / splitk;
    const int kt0 = kz * per, kt1 = min(nkt, kt0 + per);
    // The bias of this lane's column is requested HERE: behind the k-loop (where it is used, and where the barriers keep the compiler from
    // moving it up) it was a dependent round trip in front of the stores -- tools/gemm_bench.py: a launch costs 6.5 us for K = 64, 128 and 256
    // alike; what bounds these GEMMs is the fixed chain kernel arguments -> first operands -> ... -> bias -> stores -> write-back, not K.
    const int bcol = n0 + wn * 32 + (lane & 31);
    const float bv = (bias != nullptr && kz == 0 && (!TAIL || bcol < N)) ? bias[bcol] : 0.f;
    // staging: thread -> (row = tid / 4 of the tile as stored, 16 consecutive elements (tid % 4) * 16 of that row)
    const int srow = tid >> 2, sseg = (tid & 3) * 16;
    float4 ra[4][4], rb[4][4];                           // two rotating sets (long k-loops) or all four k-tiles of a short one (below)
    auto gload = [&](int set, int kt) __attribute__((always_inline)) {
        int kk = min(kt, kt1 - 1);
        if constexpr (!TAIL) {
            const float* Ap = A; const float* Bp = B;
            if (x.seg && kk >= x.seg) { kk -= x.seg; Ap = x.A2; Bp = x.B2; }      // block-uniform
            const int k0 = kk * GB_K;
            const float* pa = A_KC ? Ap + (long)(m0 + srow) * lda + k0 + sseg : Ap + (long)(k0 + srow) * lda + m0 + sseg;
            const float* pb = B_KC ? Bp + (long)(n0 + srow) * ldb + k0 + sseg : Bp + (long)(k0 + srow) * ldb + n0 + sseg;
#pragma unroll
            for (int i = 0; i < 4; ++i) { ra[set][i] = *reinterpret_cast<const float4*>(pa + 4 * i); rb[set][i] = *reinterpret_cast<const float4*>(pb + 4 * i); }
        } else {
            const int k0 = kk * GB_K;
            // operand stored [mn][k]: row = mn (always inside for A: M % 64 == 0; B rows < N), pieces along k; stored [k][mn]: row = k, pieces along mn
            const int rowa = A_KC ? m0 + srow : k0 + srow, cola = A_KC ? k0 + sseg : m0 + sseg, lima_r = A_KC ? M : K, lima_c = A_KC ? K : M;
            const int rowb = B_KC ? n0 + srow : k0 + srow, colb = B_KC ? k0 + sseg : n0 + sseg, limb_r = B_KC ? N : K, limb_c = B_KC ? K : N;
            const float* pa = A + (long)min(rowa, lima_r - 1) * lda; const float* pb = B + (long)min(rowb, limb_r - 1) * ldb;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ca = cola + 4 * i, cb = colb + 4 * i;
                const float4 va = *reinterpret_cast<const float4*>(pa + min(ca, lima_c - 4)), vb = *reinterpret_cast<const float4*>(pb + min(cb, limb_c - 4));
                ra[set][i] = (rowa < lima_r && ca + 3 < lima_c) ? va : z4; rb[set][i] = (rowb < limb_r && cb + 3 < limb_c) ? vb : z4;
            }
        }
    };
    // (Sharing the fused column sum out over the row blocks -- block `by` sums the k-tiles kt % nby == by -- was measured: the grouped
    // weight-gradient launch went from 63 to 71 us, every block then pays the final cross-lane reduction; without any column sum it is 51.)
    const bool do_colsum = !B_KC && colsum != nullptr && by == 0;
    float cs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cs[i] = 0.f;
    auto gstore = [&](int set, int buf, int kt) __attribute__((always_inline)) {
        // a k-tile past the slice (the pipeline's second half-step when the slice has an odd number of tiles) is staged as zeros: a
        // block-uniform branch -- as a multiplier on every element it was 32 of the ~90 VALU instructions a thread spends per k-tile,
        // and this kernel is bound by exactly those (tools/gemm_group_bench.py: ~500 cycles per k-tile and SIMD for 128 of MFMA)
        unsigned wa[8], wb[8], la[8], lb[8];
        // X3: the lo word of two neighbours = bf16 of what the hi word left over
        auto lo2 = [](float x, float y, unsigned h) __attribute__((always_inline)) { return gpack2(x - __uint_as_float(h << 16), y - __uint_as_float(h & 0xffff0000u)); };
        if (kt < kt1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 a = ra[set][i], b = rb[set][i];
                if constexpr (F16) {
                    wa[2 * i] = gpack2h(a.x * x.sa, a.y * x.sa); wa[2 * i + 1] = gpack2h(a.z * x.sa, a.w * x.sa);
                    wb[2 * i] = gpack2h(b.x * x.sb, b.y * x.sb); wb[2 * i + 1] = gpack2h(b.z * x.sb, b.w * x.sb);
                } else {
                    wa[2 * i] = gpack2(a.x, a.y); wa[2 * i + 1] = gpack2(a.z, a.w);
                    wb[2 * i] = gpack2(b.x, b.y); wb[2 * i + 1] = gpack2(b.z, b.w);
                }
                if constexpr (X3) {
                    la[2 * i] = lo2(a.x, a.y, wa[2 * i]); la[2 * i + 1] = lo2(a.z, a.w, wa[2 * i + 1]);
                    lb[2 * i] = lo2(b.x, b.y, wb[2 * i]); lb[2 * i + 1] = lo2(b.z, b.w, wb[2 * i + 1]);
                }
                if (do_colsum) { cs[4 * i] += b.x; cs[4 * i + 1] += b.y; cs[4 * i + 2] += b.z; cs[4 * i + 3] += b.w; }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { wa[i] = 0u; wb[i] = 0u; la[i] = 0u; lb[i] = 0u; }
        }
        const unsigned so = srow * GB_RS + sseg * 2;
        *reinterpret_cast<uint4*>(As[buf][0] + so) = make_uint4(wa[0], wa[1], wa[2], wa[3]); *reinterpret_cast<uint4*>(As[buf][0] + so + 16) = make_uint4(wa[4], wa[5], wa[6], wa[7]);
        *reinterpret_cast<uint4*>(Bs[buf][0] + so) = make_uint4(wb[0], wb[1], wb[2], wb[3]); *reinterpret_cast<uint4*>(Bs[buf][0] + so + 16) = make_uint4(wb[4], wb[5], wb[6], wb[7]);
        if constexpr (X3) {
            *reinterpret_cast<uint4*>(As[buf][NPL - 1] + so) = make_uint4(la[0], la[1], la[2], la[3]); *reinterpret_cast<uint4*>(As[buf][NPL - 1] + so + 16) = make_uint4(la[4], la[5], la[6], la[7]);
            *reinterpret_cast<uint4*>(Bs[buf][NPL - 1] + so) = make_uint4(lb[0], lb[1], lb[2], lb[3]); *reinterpret_cast<uint4*>(Bs[buf][NPL - 1] + so + 16) = make_uint4(lb[4], lb[5], lb[6], lb[7]);
        }
    };
    f32x16 acc, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
    // fragment addressing: [mn][k] rows -> lane reads 8 consecutive k of its row; [k][mn] rows -> two transpose-reads (rows
    // pr, pr + 4 of a 16-row group; the same lane map as the conv weight-gradient kernel)
    const int gi = lane & 15, gg = lane >> 4, cb = gg & 1, kh = gg >> 1;
    const unsigned colb = (16 * cb + 4 * (gi & 3)) * 2, prb = 8 * kh + (gi >> 2);
    auto frag = [&](const unsigned char* T, bool kc, int w32, int ks) __attribute__((always_inline)) -> gbf16x8 {
        if (kc) return *reinterpret_cast<const gbf16x8*>(T + (w32 * 32 + (lane & 31)) * GB_RS + (ks * 16 + 8 * (lane >> 5)) * 2);
        const unsigned p0 = 16 * ks + prb;
        const gs16x4 lo = g_lds_tr(T, p0 * GB_RS + w32 * 64 + colb), hi = g_lds_tr(T, (p0 + 4) * GB_RS + w32 * 64 + colb);
        return __builtin_bit_cast(gbf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto gmma = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const gbf16x8 a = frag(As[buf][0], A_KC, wm, ks), b = frag(Bs[buf][0], B_KC, wn, ks);
            if constexpr (X3) {                                      // small terms first; the two accumulators alternate
                const gbf16x8 al = frag(As[buf][NPL - 1], A_KC, wm, ks), bl = frag(Bs[buf][NPL - 1], B_KC, wn, ks);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc2, 0, 0, 0);
            } else if constexpr (F16) {
                const gf16x8 ah = __builtin_bit_cast(gf16x8, a), bh = __builtin_bit_cast(gf16x8, b);
                if (ks & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            } else if (ks & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    };
    if (kt1 - kt0 >= 3 && kt1 - kt0 <= 4) {
        // Short k-loops (K = 256 per slice: every Dense / GRU-projection GEMM of the model at hidden_dim 256): ALL k-tiles are requested at
        // once -- one round trip.  The rotating form below keeps two tiles in flight, so a slice of four tiles pays the latency of an operand
        // that the previous kernel left behind the fabric (~2 800 cycles: nothing a launch reads first is in its XCD's L2) twice over.
        gload(0, kt0); gload(1, kt0 + 1); gload(2, kt0 + 2); gload(3, kt0 + 3);   // (tiles past the slice: clamped address, staged as zeros)
        gstore(0, 0, kt0);
        __syncthreads();
        gmma(0); gstore(1, 1, kt0 + 1);
        __syncthreads();
        gmma(1); gstore(2, 0, kt0 + 2);
        __syncthreads();
        gmma(0); gstore(3, 1, kt0 + 3);
        __syncthreads();
        gmma(1);
    } else if (kt0 < kt1) {
        gload(0, kt0); gload(1, kt0 + 1);
        gstore(0, 0, kt0); gload(0, kt0 + 2);
        __syncthreads();
        for (int kt = kt0; kt < kt1; kt += 2) {
            gmma(0); gstore(1, 1, kt + 1); gload(1, kt + 3);
            __syncthreads();
            gmma(1); gstore(0, 0, kt + 2); gload(0, kt + 4);
            __syncthreads();
        }
    }
    const float unscale = F16 ? 1.f / (x.sa * x.sb) : 1.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = F16 ? (acc[i] + acc2[i]) * unscale : acc[i] + acc2[i];
    if (do_colsum) {                                  // threads with equal tid % 4 hold the same 16 columns
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = cs[i];
            v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lane < 4) csum[wave][lane * 16 + i] = v;
        }
        __syncthreads();
        if (tid < 64 && (!TAIL || n0 + tid < N)) {
            const float v = csum[0][tid] + csum[1][tid] + csum[2][tid] + csum[3][tid];
            atomicAdd(colsum + n0 + tid, v);
        }
    }
    const int col = n0 + wn * 32 + (lane & 31);
    if (TAIL && col >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (TAIL && row >= M) continue;
        float v = acc[r] + bv;
        float* dst = C + (long)row * ldc + col;
        if (splitk > 1) {
            atomicAdd(dst, v);
        } else {
            if (act == ACT_TANH) v = fast_tanh(v);
            else if (act == ACT_RELU) v = fmaxf(v, 0.f);
            if (accumulate) v += *dst;
            *dst = v;
        }
    }
}

// XCD-aware tile order.  Workgroups go to the eight XCDs round-robin by their linear id, and every XCD has its own 4 MB L2: in
// dispatch order the tiles that share operand rows (the N / 64 column tiles of one row block; the tiles of one split-K slice) sit
// on eight different L2s, each of which fetches those rows from the fabric again -- the grouped weight gradient moved 300 MB for
// 50 MB of operands and ran at the 4.7 TB/s that comes to (91 TFLOP/s).  Here XCD c takes the c-th CONTIGUOUS eighth of the tile
// sequence (column tile fastest, then row block, then split-K slice / batch), so neighbours in that sequence share an L2.
__device__ __forceinline__ unsigned xcd_order(unsigned l, unsigned total) {
    const unsigned chunk = total >> 3, body = chunk << 3;           // the last total % 8 blocks keep their place
    return l < body ? (l & 7u) * chunk + (l >> 3) : l;
}

// (X3: 73 KB of LDS -- past the 64 KB a static allocation may have: the launcher passes it as dynamic LDS)
extern __shared__ __attribute__((aligned(16))) unsigned char gemm_dyn_lds[];
template <bool X3> __device__ __forceinline__ GemmLdsT<X3>* gemm_lds() {
    if constexpr (X3) return reinterpret_cast<GemmLdsT<true>*>(gemm_dyn_lds);
    else { __shared__ __attribute__((aligned(16))) GemmLdsT<false> L; return &L; }
}
template <bool A_KC, bool B_KC, bool TAIL = false, bool X3 = false, bool F16 = false>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmBatch gb, int M, int N,
                                                        int K, int lda, int ldb, int ldc, int act, int accumulate, int splitk, const GemmX x) {
    const unsigned nbx = gridDim.x, nby = gridDim.y;
    const unsigned t = xcd_order(blockIdx.x + nbx * (blockIdx.y + nby * blockIdx.z), nbx * nby * gridDim.z);
    const int bx = (int)(t % nbx), by = (int)(t / nbx % nby), z = (int)(t / (nbx * nby));
    const int bz = z / splitk, kz = z - bz * splitk;
    GemmLdsT<X3>* L = gemm_lds<X3>();
    gemm_bf16_tile<A_KC, B_KC, TAIL, X3, F16>(gb.A[bz], gb.B[bz], gb.C[bz], gb.bias[bz], gb.colsum[bz], M, N, K, lda, ldb, ldc, act, accumulate, splitk,
                                              bx, by, kz, L, x);
}

// Weight gradients of MANY Dense / GRU kernels in one launch (they are off the critical path of the backward pass: the host collects
// them and launches the group once, behind the last input gradient): dW_i[in_i, out_i] += X_i^T dY_i, db_i += column sums of dY_i, the
// problems of any (bf16-path) shape.  A block finds its problem in a table sorted by first block (one load round trip, as in
// pack_weights_kernel), then runs the same tile code as sar_gemm_wgrad_f32.  Launches of 7-35 us that each filled a fraction of
// the chip become one.
struct WgRow { const float* X; const float* dY; float* dW; float* db; int M, N, K, lda, ldb, ldc, splitk, first_block; };   // 64 bytes
template <bool X3, bool F16 = false>
__global__ __launch_bounds__(256) void gemm_bf16_wgrad_group_kernel(const WgRow* __restrict__ tab, int n, float gscale) {

    GemmLdsT<X3>* L = gemm_lds<X3>();
    int r = -1;
    for (int base = 0; base < n; base += 64) {
        const int q = base + (threadIdx.x & 63);
        r += __popcll(__ballot(q < n && tab[q].first_block <= (int)blockIdx.x));
    }
    r = __builtin_amdgcn_readfirstlane(r);
    WgRow p = tab[r];
    // F16: the dY operand times the gradient scale, lowered by the problem's own 2^-shift (bits 8.. of splitk; the plan's pad_): a per-utterance
    // layer's gradient rows are ~L times a per-frame layer's, and BatchNorm backward can multiply them by rstd <= 31 on top
    const GemmX x{1.f, gscale * __builtin_amdgcn_exp2f(-(float)(p.splitk >> 8)), nullptr, nullptr, 0};
    p.splitk &= 255;
    const int nbx = (p.N + 63) / 64, nby = (p.M + 63) / 64;
    const int local = (int)xcd_order(blockIdx.x - p.first_block, (unsigned)(nbx * nby * p.splitk));   // first_block % 8 == 0 (the plan pads)
    const int bx = local % nbx, by = (local / nbx) % nby, kz = local / (nbx * nby);
    if (kz >= p.splitk) return;                                      // padding blocks of the plan
    if ((p.M | p.N | p.K) & 63) gemm_bf16_tile<false, false, true, X3, F16>(p.X, p.dY, p.dW, nullptr, p.db, p.M, p.N, p.K, p.lda, p.ldb, p.ldc, ACT_NONE, 1, p.splitk, bx, by, kz, L, x);
    else gemm_bf16_tile<false, false, false, X3, F16>(p.X, p.dY, p.dW, nullptr, p.db, p.M, p.N, p.K, p.lda, p.ldb, p.ldc, ACT_NONE, 1, p.splitk, bx, by, kz, L, x);
}

}  // namespace

// GEMM arithmetic: 0 = fp32 MFMA (default), 1 = bf16 MFMA with fp32 accumulate for 64-aligned problems (the "bf16" mode of
// BASELINE configs[1]; SAR_GEMM_MATH=bf16), 2 = the same kernels in split precision (three bf16 MFMAs per product block; the
// fp32-class "bf16x3" mode; SAR_GEMM_MATH=bf16x3).  Host-side, process-wide.
static int g_gemm_math = !getenv("SAR_GEMM_MATH") ? 0 : !strcmp(getenv("SAR_GEMM_MATH"), "bf16") ? 1 : !strcmp(getenv("SAR_GEMM_MATH"), "bf16x3") ? 2 : 0;
extern "C" int sar_set_gemm_math(int mode) { if (mode < 0 || mode > 2) return SAR_EINVAL; g_gemm_math = mode; return SAR_OK; }
// launch of an X3 instantiation: its LDS is dynamic and past 64 KB (raised once per device)
template <typename KT, typename... Args>
static int launch_x3(KT kern, PerDeviceFlag& flag, dim3 grid, hipStream_t stream, Args... args) {
    const size_t lds = sizeof(GemmLdsT<true>);
    if (!flag) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SAR_ELAUNCH;
        flag = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, args...);
    return SAR_OK;
}
extern "C" int sar_get_gemm_math(void) { return g_gemm_math; }

// C-ABI: see include/sarnet.h
static int gemm_launch(const GemmBatch& gb, int nb, int M, int N, int K, int lda, int ldb,
                       int ldc, int a_kc, int b_kc, int act, int accumulate, int splitk, hipStream_t stream) {
    if (nb < 1 || nb > 4 || M <= 0 || N <= 0 || K <= 0 || splitk < 1) return SAR_EINVAL;
    if (splitk > 1 && act != ACT_NONE) return SAR_EINVAL;  // split-K accumulates atomically: C must be pre-zeroed (or hold the addend)
    bool aligned = true;
    for (int i = 0; i < nb; ++i) {
        if (!gb.A[i] || !gb.B[i] || !gb.C[i]) return SAR_EINVAL;
        aligned = aligned && (reinterpret_cast<uintptr_t>(gb.A[i]) & 15) == 0 && (reinterpret_cast<uintptr_t>(gb.B[i]) & 15) == 0;
    }
    if (nb > 1 && !aligned) return SAR_EINVAL;             // a batch runs one code path: all operands 16-byte aligned
    dim3 grid(cdiv(N, BN), cdiv(M, BM), splitk * nb), block(256);
    // bf16 MFMA path: whole 64 x 64 x 64 tiles, 16-byte aligned rows (every Dense / GRU-projection GEMM of the model)
    // bf16 MFMA path with tails: extents and row strides that are multiples of 4, from 64 up (smaller problems keep the fp32 kernel)
    const bool pieces = ((M | N | K | lda | ldb) & 3) == 0 && aligned;
    const bool whole = pieces && M % 64 == 0 && N % 64 == 0 && K % 64 == 0;
    const bool tail16 = g_gemm_math >= 1 && pieces && !whole && M >= 64 && N >= 64 && K >= 64 && (K + 63) / 64 >= splitk;
    const bool x3 = g_gemm_math == 2;
    if (tail16) {
        grid.x = cdiv(N, 64); grid.y = cdiv(M, 64);
#define LAUNCH16T(AK, BK_)                                                                                           \
    do { if (x3) { static PerDeviceFlag fl; const int rc = launch_x3(gemm_bf16_kernel<AK, BK_, true, true>, fl, grid, stream, gb, M, N, K, lda, ldb, ldc, act, accumulate, splitk, GX0); if (rc != SAR_OK) return rc; } \
         else hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK_, true>), grid, block, 0, stream, gb, M, N, K, lda, ldb, ldc, act, accumulate, splitk, GX0); } while (0)
        if (a_kc && b_kc) LAUNCH16T(true, true);
        else if (a_kc && !b_kc) LAUNCH16T(true, false);
        else if (!a_kc && b_kc) LAUNCH16T(false, true);
        else LAUNCH16T(false, false);
#undef LAUNCH16T
        SAR_CHECK_LAUNCH();
        return SAR_OK;
    }
    if (g_gemm_math >= 1 && M % 64 == 0 && N % 64 == 0 && K % 64 == 0 && (lda & 3) == 0 && (ldb & 3) == 0 && aligned && K / 64 >= splitk) {
#define LAUNCH16(AK, BK_)                                                                                            \
    do { if (x3) { static PerDeviceFlag fl; const int rc = launch_x3(gemm_bf16_kernel<AK, BK_, false, true>, fl, grid, stream, gb, M, N, K, lda, ldb, ldc, act, accumulate, splitk, GX0); if (rc != SAR_OK) return rc; } \
         else hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK_>), grid, block, 0, stream, gb, M, N, K, lda, ldb, ldc, act, accumulate, splitk, GX0); } while (0)
        if (a_kc && b_kc) LAUNCH16(true, true);
        else if (a_kc && !b_kc) LAUNCH16(true, false);
        else if (!a_kc && b_kc) LAUNCH16(false, true);
        else LAUNCH16(false, false);
#undef LAUNCH16
        SAR_CHECK_LAUNCH();
        return SAR_OK;
    }
#define LAUNCH(AK, BK_)                                                                                              \
    hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_>), grid, block, 0, stream, gb, M, N, K, lda, ldb, ldc, act, accumulate, splitk)
    if (a_kc && b_kc) LAUNCH(true, true);
    else if (a_kc && !b_kc) LAUNCH(true, false);
    else if (!a_kc && b_kc) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                            int ldb, int ldc, int a_kc, int b_kc, int act, int accumulate, int splitk,
                            hipStream_t stream) {
    GemmBatch gb{};
    gb.A[0] = A; gb.B[0] = B; gb.C[0] = C; gb.bias[0] = bias;
    return gemm_launch(gb, 1, M, N, K, lda, ldb, ldc, a_kc, b_kc, act, accumulate, splitk, stream);
}

// include/sarnet.h.  C (M,N) (+)= A B + A2 B2 in ONE launch on the 2-byte MFMA kernels: A, A2 (M, Kseg) row-major (lda); B, B2 stored (N, Kseg)
// (b_kc = 1) or (Kseg, N) (b_kc = 0), ldb; A2 / B2 NULL: one product.  scale_a > 0: single-pass IEEE fp16 operands, A times scale_a and B
// times scale_b (powers of two) while staged, the result times 1 / (scale_a scale_b) -- the backward products of the split-precision mode;
// scale_a == 0: the process-wide GEMM arithmetic (bf16 or bf16x3).  Whole 64 x 64 x 64 tiles and 16-byte aligned operands only, GEMM math
// bf16 / bf16x3 (else SAR_EINVAL: the caller runs sar_gemm_f32 per product).
extern "C" int sar_gemm_pair_f32(const float* A, const float* B, const float* A2, const float* B2, float* C, int M, int N, int Kseg, int lda,
                                 int ldb, int ldc, int b_kc, int accumulate, float scale_a, float scale_b, hipStream_t stream) {
    if (!A || !B || !C || (!A2) != (!B2) || M < 64 || N < 64 || Kseg < 64 || g_gemm_math < 1) return SAR_EINVAL;
    if ((M | N | Kseg) & 63 || (lda | ldb) & 3 || scale_a < 0.f || (scale_a > 0.f && !(scale_b > 0.f))) return SAR_EINVAL;
    const uintptr_t al = reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(A2) | reinterpret_cast<uintptr_t>(B2);
    if (al & 15) return SAR_EINVAL;
    GemmBatch gb{};
    gb.A[0] = A; gb.B[0] = B; gb.C[0] = C;
    const GemmX x{scale_a > 0.f ? scale_a : 1.f, scale_a > 0.f ? scale_b : 1.f, A2, B2, A2 ? Kseg / 64 : 0};
    const int K = A2 ? 2 * Kseg : Kseg;
    dim3 grid(N / 64, M / 64, 1), block(256);
    if (scale_a > 0.f) {
        if (b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, false, false, true>), grid, block, 0, stream, gb, M, N, K, lda, ldb, ldc, (int)ACT_NONE, accumulate, 1, x);
        else hipLaunchKernelGGL((gemm_bf16_kernel<true, false, false, false, true>), grid, block, 0, stream, gb, M, N, K, lda, ldb, ldc, (int)ACT_NONE, accumulate, 1, x);
    } else if (g_gemm_math == 2) {
        static PerDeviceFlag f1, f0;
        const int rc = b_kc ? launch_x3(gemm_bf16_kernel<true, true, false, true>, f1, grid, stream, gb, M, N, K, lda, ldb, ldc, (int)ACT_NONE, accumulate, 1, x)
                            : launch_x3(gemm_bf16_kernel<true, false, false, true>, f0, grid, stream, gb, M, N, K, lda, ldb, ldc, (int)ACT_NONE, accumulate, 1, x);
        if (rc != SAR_OK) return rc;
    } else {
        if (b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, stream, gb, M, N, K, lda, ldb, ldc, (int)ACT_NONE, accumulate, 1, x);
        else hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, 0, stream, gb, M, N, K, lda, ldb, ldc, (int)ACT_NONE, accumulate, 1, x);
    }
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// Weight gradient of a Dense layer with its bias gradient: dW[in,out] += X^T[in,R] dY[R,out], db[out] += sum_R dY.
// X is (R, in) row-major (lda), dY is (R, out) row-major (ldb); both results ACCUMULATE (split-K uses atomics).
extern "C" int sar_gemm_wgrad_f32(const float* X, const float* dY, float* dW, float* db, int in_dim, int out_dim, int R,
                                  int lda, int ldb, int ldc, int splitk, hipStream_t stream) {
    GemmBatch gb{};
    gb.A[0] = X; gb.B[0] = dY; gb.C[0] = dW; gb.colsum[0] = db;
    return gemm_launch(gb, 1, in_dim, out_dim, R, lda, ldb, ldc, 0, 0, ACT_NONE, 1, splitk, stream);
}

// include/sarnet.h: sarGemmBatch (host struct of device pointers; entries >= nbatch ignored)
struct sarGemmBatch { const float* A[4]; const float* B[4]; float* C[4]; const float* bias[4]; };

// nbatch (1..4) problems of equal shape, layouts and flags in ONE launch: C[i] = act(op(A[i]) op(B[i]) + bias[i]) as sar_gemm_f32.
extern "C" int sar_gemm_batched_f32(const sarGemmBatch* b, int nbatch, int M, int N, int K, int lda, int ldb, int ldc, int a_kc,
                                    int b_kc, int act, int accumulate, int splitk, hipStream_t stream) {
    if (!b || nbatch < 1 || nbatch > 4) return SAR_EINVAL;
    GemmBatch gb{};
    for (int i = 0; i < nbatch; ++i) { gb.A[i] = b->A[i]; gb.B[i] = b->B[i]; gb.C[i] = b->C[i]; gb.bias[i] = b->bias[i]; }
    return gemm_launch(gb, nbatch, M, N, K, lda, ldb, ldc, a_kc, b_kc, act, accumulate, splitk, stream);
}

// nbatch weight gradients of equal shape in one launch: A[i] = X_i (R, in), B[i] = dY_i (R, out), C[i] = dW_i (+=), bias[i] = db_i
// (+=, nullable), as sar_gemm_wgrad_f32.
extern "C" int sar_gemm_wgrad_batched_f32(const sarGemmBatch* b, int nbatch, int in_dim, int out_dim, int R, int lda, int ldb,
                                          int ldc, int splitk, hipStream_t stream) {
    if (!b || nbatch < 1 || nbatch > 4) return SAR_EINVAL;
    GemmBatch gb{};
    for (int i = 0; i < nbatch; ++i) { gb.A[i] = b->A[i]; gb.B[i] = b->B[i]; gb.C[i] = b->C[i]; gb.colsum[i] = const_cast<float*>(b->bias[i]); }
    return gemm_launch(gb, nbatch, in_dim, out_dim, R, lda, ldb, ldc, 0, 0, ACT_NONE, 1, splitk, stream);
}

// include/sarnet.h: sarGemmWgrad.  Host-only planning: validates that every problem runs on the bf16-MFMA path (SAR_EINVAL otherwise:
// the caller then launches them one by one) and writes the device table rows (64 bytes each) + the grid size.
struct sarGemmWgrad { const float* X; const float* dY; float* dW; float* db; int in_dim, out_dim, R, lda, ldb, ldc, splitk, pad_; };
extern "C" int sar_gemm_wgrad_group_plan(const sarGemmWgrad* probs, int n, void* table_out, int* total_blocks) {
    if (!probs || n < 1 || n > 4096 || !table_out || !total_blocks || g_gemm_math < 1) return SAR_EINVAL;
    WgRow* rows = static_cast<WgRow*>(table_out);
    long blk = 0;
    for (int i = 0; i < n; ++i) {
        const sarGemmWgrad& q = probs[i];
        const int M = q.in_dim, N = q.out_dim, K = q.R;
        if (!q.X || !q.dY || !q.dW || M < 64 || N < 64 || K < 64 || q.splitk < 1 || q.splitk > 255 || (K + 63) / 64 < q.splitk || q.pad_ < 0 || q.pad_ > 24) return SAR_EINVAL;
        if (((M | N | K | q.lda | q.ldb) & 3) || ((reinterpret_cast<uintptr_t>(q.X) | reinterpret_cast<uintptr_t>(q.dY)) & 15)) return SAR_EINVAL;
        WgRow r;
        r.X = q.X; r.dY = q.dY; r.dW = q.dW; r.db = q.db; r.M = M; r.N = N; r.K = K; r.lda = q.lda; r.ldb = q.ldb; r.ldc = q.ldc;
        r.splitk = q.splitk | (q.pad_ << 8); r.first_block = (int)blk;   // pad_ (sar_gemm_wgrad_group_scaled_f32): this problem's gradient scale is gscale * 2^-pad_
        rows[i] = r;
        blk += ((long)((N + 63) / 64) * ((M + 63) / 64) * q.splitk + 7) & ~7L;   // every problem starts on XCD 0 (xcd_order); <= 7 idle blocks each
        if (blk > (1L << 30)) return SAR_EINVAL;
    }
    *total_blocks = (int)blk;
    return SAR_OK;
}
extern "C" int sar_gemm_wgrad_group_f32(const void* table_dev, int n, int total_blocks, hipStream_t stream) {
    if (!table_dev || n < 1 || total_blocks < 1 || g_gemm_math < 1) return SAR_EINVAL;
    if (g_gemm_math == 2) {
        static PerDeviceFlag fl;
        const int rc = launch_x3(gemm_bf16_wgrad_group_kernel<true>, fl, dim3(total_blocks), stream, static_cast<const WgRow*>(table_dev), n, 1.f);
        if (rc != SAR_OK) return rc;
    } else {
        hipLaunchKernelGGL(gemm_bf16_wgrad_group_kernel<false>, dim3(total_blocks), dim3(256), 0, stream, static_cast<const WgRow*>(table_dev), n, 1.f);
    }
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
// gscale > 0: the group in single-pass fp16 (gemm_bf16_tile: F16), every dY operand times gscale while staged; 0: the GEMM arithmetic in force
extern "C" int sar_gemm_wgrad_group_scaled_f32(const void* table_dev, int n, int total_blocks, float gscale, hipStream_t stream) {
    if (!table_dev || n < 1 || total_blocks < 1 || g_gemm_math < 1 || gscale < 0.f) return SAR_EINVAL;
    if (gscale > 0.f) {
        hipLaunchKernelGGL((gemm_bf16_wgrad_group_kernel<false, true>), dim3(total_blocks), dim3(256), 0, stream, static_cast<const WgRow*>(table_dev), n, gscale);
        SAR_CHECK_LAUNCH();
        return SAR_OK;
    }
    return sar_gemm_wgrad_group_f32(table_dev, n, total_blocks, stream);
}

// ---------------------------------------------------------------------------------------------------- Dense(tanh) + LayerNormalization in one launch
// DS(hidden_dim, 'tanh') -> LN, the reference's per-frame pairs (model.py:253-254 CNN_LIN, 265-266 CTC_DS, 279-280 AR_DS): with 64 x 64
// output tiles a LayerNorm row (256 units) is spread over four workgroups, so the normalisation was a launch of its own behind the GEMM -- 12 + 5 and
// 14 + 4.4 us in the step for 0.3 GFLOP each.  Here a workgroup owns 16 COMPLETE rows: wave w multiplies columns [64 w, 64 w + 64) as four
// v_mfma_f32_16x16x32_bf16 tiles in split precision (hi x hi + hi x lo + lo x hi, the GEMM arithmetic "bf16x3"), operands straight from global memory
// into fragments (an x row piece is 8 consecutive floats of the lane's row; a kernel piece is a float4 of FOUR consecutive columns at one k, whose
// components go to the four tiles: tile q of the wave holds columns 64 w + 4 (lane % 16) + q, so a lane ends up with four consecutive columns of its
// four rows -- 16-byte stores), two k-steps of 32 in flight; bias, tanh, the row's mean and variance (16-lane DPP sums, the four waves meet in LDS
// twice) and the normalisation follow in registers.  Writes both tensors the backward reads: tanh's output (the LayerNorm's input) and the result.
namespace {
constexpr int DL_N = 256, DL_ROWS = 16, DL_RS = DL_N + 4;   // (DL_RS: row stride of the k-groups' partial tiles in LDS, floats)
struct DlStage { float4 a0, a1; float4 w[8]; };

// KG k-groups of four waves: group g multiplies k in [g K / KG, (g + 1) K / KG) (a multiple of 64 each) and the groups' tiles meet in LDS -- 52
// workgroups on 52 CUs are all this launch is at the step's 832 rows, each wave a chain of load round trips: more waves per CU is what shortens it.
template <int KG>
__global__ __launch_bounds__(256 * KG) void dense_tanh_ln_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ act_out,
                                                                 float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int R, int K, float eps) {
    __shared__ float part[2][4][DL_ROWS];
    __shared__ __attribute__((aligned(16))) float red[KG > 1 ? KG - 1 : 1][KG > 1 ? DL_ROWS : 1][KG > 1 ? DL_RS : 4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, w = wv & 3, g = wv >> 2, r16 = lane & 15, kg = lane >> 4;
    const int row0 = blockIdx.x * DL_ROWS;
    const int arow = min(row0 + r16, R - 1);                          // (rows past R: a clamped read, no store)
    const int Kg = K / KG, k0 = g * Kg;
    const float* xa = x + (size_t)arow * K + k0 + 8 * kg;
    const float* wb = W + (size_t)(k0 + 8 * kg) * DL_N + 64 * w + 4 * r16;
    auto load = [&](DlStage& s, int kk) __attribute__((always_inline)) {
        s.a0 = *reinterpret_cast<const float4*>(xa + kk); s.a1 = *reinterpret_cast<const float4*>(xa + kk + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) s.w[i] = *reinterpret_cast<const float4*>(wb + (size_t)(kk + i) * DL_N);
    };
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const DlStage& s) __attribute__((always_inline)) {
        const float av[8] = {s.a0.x, s.a0.y, s.a0.z, s.a0.w, s.a1.x, s.a1.y, s.a1.z, s.a1.w};
        gbf16x8 ah, al;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const __bf16 h = (__bf16)av[i]; ah[i] = h; al[i] = (__bf16)(av[i] - (float)h); }
        float wq[4][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { wq[0][i] = s.w[i].x; wq[1][i] = s.w[i].y; wq[2][i] = s.w[i].z; wq[3][i] = s.w[i].w; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            gbf16x8 bh, bl;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const __bf16 h = (__bf16)wq[q][i]; bh[i] = h; bl[i] = (__bf16)(wq[q][i] - (float)h); }
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[q], 0, 0, 0);      // small terms first
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[q], 0, 0, 0);
        }
    };
    DlStage s0, s1;
    load(s0, 0); load(s1, 32);
    for (int kk = 0; kk < Kg; kk += 64) {                             // Kg % 64 == 0 (the launcher picks KG so)
        compute(s0);
        if (kk + 64 < Kg) load(s0, kk + 64);
        compute(s1);
        if (kk + 96 < Kg) load(s1, kk + 96);
    }
    // lane: rows 4 kg + i (i < 4), columns c0 .. c0 + 3
    const int c0 = 64 * w + 4 * r16;
    if constexpr (KG > 1) {
        if (g > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&red[g - 1][4 * kg + i][c0]) = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
        }
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < KG - 1; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 v = *reinterpret_cast<const float4*>(&red[j][4 * kg + i][c0]);
                    acc[0][i] += v.x; acc[1][i] += v.y; acc[2][i] += v.z; acc[3][i] += v.w;
                }
        }
    }
    const bool own = g == 0;                                          // (block-uniform per wave) the first group finishes the rows; the others keep the barriers company
    float t[4][4], rs[4], mu[4], d[4][4];
    if (own) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + c0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            t[i][0] = fast_tanh(acc[0][i] + bv.x); t[i][1] = fast_tanh(acc[1][i] + bv.y); t[i][2] = fast_tanh(acc[2][i] + bv.z); t[i][3] = fast_tanh(acc[3][i] + bv.w);
            float s = (t[i][0] + t[i][1]) + (t[i][2] + t[i][3]);
            s += SAR_DPP_F(s, 0xB1); s += SAR_DPP_F(s, 0x4E); s += SAR_DPP_F(s, 0x141); s += SAR_DPP_F(s, 0x140);   // the 16 lanes of this k-group = the wave's 64 columns
            rs[i] = s;
        }
        if (r16 == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) part[0][w][4 * kg + i] = rs[i];
        }
    }
    __syncthreads();
    if (own) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = 4 * kg + i;
            mu[i] = ((part[0][0][rr] + part[0][1][rr]) + (part[0][2][rr] + part[0][3][rr])) * (1.f / DL_N);
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { d[i][q] = t[i][q] - mu[i]; s = fmaf(d[i][q], d[i][q], s); }
            s += SAR_DPP_F(s, 0xB1); s += SAR_DPP_F(s, 0x4E); s += SAR_DPP_F(s, 0x141); s += SAR_DPP_F(s, 0x140);
            rs[i] = s;
        }
        if (r16 == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) part[1][w][4 * kg + i] = rs[i];
        }
    }
    __syncthreads();
    if (!own) return;
    const float4 gv = *reinterpret_cast<const float4*>(gamma + c0), be = *reinterpret_cast<const float4*>(beta + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = 4 * kg + i, row = row0 + rr;
        const float var = ((part[1][0][rr] + part[1][1][rr]) + (part[1][2][rr] + part[1][3][rr])) * (1.f / DL_N);
        const float r = 1.f / sqrtf(var + eps);
        if (row < R) {
            *reinterpret_cast<float4*>(act_out + (size_t)row * DL_N + c0) = make_float4(t[i][0], t[i][1], t[i][2], t[i][3]);
            *reinterpret_cast<float4*>(y + (size_t)row * DL_N + c0) = make_float4(fmaf(d[i][0] * r, gv.x, be.x), fmaf(d[i][1] * r, gv.y, be.y),
                                                                                  fmaf(d[i][2] * r, gv.z, be.z), fmaf(d[i][3] * r, gv.w, be.w));
            if (w == 0 && r16 == 0) { mean[row] = mu[i]; rstd[row] = r; }
        }
    }
}
}  // namespace

// include/sarnet.h.  N = 256 only, K a multiple of 64, 16-byte aligned x (R, K) and W (K, 256), GEMM arithmetic bf16x3 -- otherwise SAR_EINVAL
// (the caller runs sar_gemm_f32 + sar_layernorm_fwd_f32).
extern "C" int sar_dense_tanh_ln_f32(const float* x, const float* W, const float* bias, const float* gamma, const float* beta, float* act_out, float* y,
                                     float* mean, float* rstd, long R, int K, int N, float eps, hipStream_t stream) {
    if (!x || !W || !bias || !gamma || !beta || !act_out || !y || !mean || !rstd || R < 1 || R > (1L << 30) || N != DL_N || K < 64 || K % 64 || g_gemm_math != 2)
        return SAR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(gamma) |
         reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(act_out) | reinterpret_cast<uintptr_t>(y)) & 15)
        return SAR_EINVAL;
    static const int kg_env = getenv("SAR_DENSE_LN_KG") ? atoi(getenv("SAR_DENSE_LN_KG")) : 4;
    const dim3 grid(cdiv(R, DL_ROWS));
    const int kgs = kg_env >= 4 && K % 256 == 0 ? 4 : kg_env >= 2 && K % 128 == 0 ? 2 : 1;
    if (kgs == 4) hipLaunchKernelGGL(dense_tanh_ln_kernel<4>, grid, dim3(1024), 0, stream, x, W, bias, gamma, beta, act_out, y, mean, rstd, (int)R, K, eps);
    else if (kgs == 2) hipLaunchKernelGGL(dense_tanh_ln_kernel<2>, grid, dim3(512), 0, stream, x, W, bias, gamma, beta, act_out, y, mean, rstd, (int)R, K, eps);
    else hipLaunchKernelGGL(dense_tanh_ln_kernel<1>, grid, dim3(256), 0, stream, x, W, bias, gamma, beta, act_out, y, mean, rstd, (int)R, K, eps);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
