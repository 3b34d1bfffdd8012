// VladPooling.call (VLAD.py:26-49) on gfx950: NetVLAD / GhostVLAD soft-assignment residual pooling.
// One workgroup per utterance; the (L, K+G) assignment matrix is staged in LDS; the reference's
// B x L x (K+G) x D broadcast tensor (VLAD.py:40-41) is never materialised:
//     V[k,d] = sum_l A[l,k] x[l,d]  -  (sum_l A[l,k]) c[k,d]        (only the K non-ghost clusters)
//     out[k,:] = V[k,:] * rsqrt(max(|V[k,:]|^2, 1e-12))               (tf.nn.l2_normalize, VLAD.py:47)
// Saved for backward: A (B,L,K+G), inv (B,K) (negative sign flags the clamped branch).
#include "common.h"

namespace {

// In-kernel phase timing (tools/vlad_timing.py; only with -DSAR_TIMING into tools/libsarnet_timing_vlad.so, never part of libsarnet_hip.so):
// thread 0 of every block stamps s_memtime (shader clock) / s_memrealtime (100 MHz, device-wide: slots 0, 15).
#ifdef SAR_TIMING
__device__ unsigned long long* g_tbuf_vlad = nullptr;
#define TSTAMP(slot)                                                                                                   \
    do {                                                                                                               \
        if (g_tbuf_vlad && threadIdx.x == 0)                                                                           \
            g_tbuf_vlad[((long)blockIdx.x + (long)gridDim.x * blockIdx.y) * 16 + (slot)] = ((slot) == 0 || (slot) == 15) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define TSTAMP(slot) do {} while (0)
#endif

constexpr int KC = 16;      // clusters accumulated per register pass
constexpr int XB = 13;      // feature rows fetched per batch (independent loads in flight)

// Stage the (L, K+G) score tile into LDS with coalesced loads, soft-max each row in LDS (one thread per row; rows are
// short), then write the assignment matrix back coalesced.  (v1 read the scores with per-thread serial global loads:
// ~100 dependent L2 round trips per row.)
__device__ __forceinline__ void softmax_rows_to_lds(const float* __restrict__ score, float* __restrict__ A_s,
                                                    float* __restrict__ A_out, int L, int KG) {
    for (int i = threadIdx.x; i < L * KG; i += blockDim.x) A_s[i] = score[i];
    __syncthreads();
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        float* s = A_s + l * KG;
        float m = -INFINITY;
        for (int k = 0; k < KG; ++k) m = fmaxf(m, s[k]);
        float sum = 0.f;
        for (int k = 0; k < KG; ++k) { const float e = expf(s[k] - m); s[k] = e; sum += e; }
        const float inv = 1.f / sum;
        for (int k = 0; k < KG; ++k) s[k] *= inv;
    }
    __syncthreads();
    if (A_out) for (int i = threadIdx.x; i < L * KG; i += blockDim.x) A_out[i] = A_s[i];
}

__global__ __launch_bounds__(256) void vlad_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ score,
                                                       const float* __restrict__ centers, float* __restrict__ out,
                                                       float* __restrict__ A_out, float* __restrict__ inv_out, int L,
                                                       int D, int K, int KG) {
    extern __shared__ float sm[];
    float* A_s = sm;                   // [L][KG]
    float* asum = A_s + L * KG;        // [KG]
    float* ss = asum + KG;             // [K]
    float* res_s = ss + K;             // [K][D] un-normalised residuals (written once to HBM after the norms are known)
    const int b = blockIdx.x;
    const float* x = feat + (long)b * L * D;
    softmax_rows_to_lds(score + (long)b * L * KG, A_s, A_out + (long)b * L * KG, L, KG);
    for (int k = threadIdx.x; k < K; k += blockDim.x) ss[k] = 0.f;
    __syncthreads();
    for (int k = threadIdx.x; k < KG; k += blockDim.x) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += A_s[l * KG + k];
        asum[k] = s;
    }
    __syncthreads();
    float* o = out + (long)b * K * D;
    for (int k0 = 0; k0 < K; k0 += KC) {
        for (int d0 = 0; d0 < D; d0 += blockDim.x) {           // wave-uniform trip count (shuffles below)
            const int d = d0 + threadIdx.x;
            const bool d_ok = d < D;
            float acc[KC];
#pragma unroll
            for (int i = 0; i < KC; ++i) acc[i] = 0.f;
            for (int l0 = 0; l0 < L; l0 += XB) {                 // XB independent loads in flight per batch
                float xv[XB];
#pragma unroll
                for (int j = 0; j < XB; ++j) xv[j] = (d_ok && l0 + j < L) ? x[(long)(l0 + j) * D + d] : 0.f;
#pragma unroll
                for (int j = 0; j < XB; ++j)
                    if (l0 + j < L) {
#pragma unroll
                        for (int i = 0; i < KC; ++i)
                            if (k0 + i < K) acc[i] = fmaf(A_s[(l0 + j) * KG + k0 + i], xv[j], acc[i]);
                    }
            }
#pragma unroll
            for (int i = 0; i < KC; ++i)
                if (d_ok && k0 + i < K) {
                    const float r = acc[i] - asum[k0 + i] * centers[(long)(k0 + i) * D + d];
                    res_s[(k0 + i) * D + d] = r;
                    acc[i] = r * r;
                } else acc[i] = 0.f;
            // one LDS atomic per wave and cluster (256 same-address float atomics per cluster serialise badly)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int i = 0; i < KC; ++i) acc[i] += __shfl_xor(acc[i], off, 64);
            if ((threadIdx.x & 63) == 0)
#pragma unroll
                for (int i = 0; i < KC; ++i)
                    if (k0 + i < K) atomicAdd(&ss[k0 + i], acc[i]);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float s = ss[k];
        const float iv = 1.f / sqrtf(fmaxf(s, 1e-12f));
        inv_out[(long)b * K + k] = s >= 1e-12f ? iv : -iv;
    }
    for (int i = threadIdx.x; i < K * D; i += blockDim.x) {
        const float iv = 1.f / sqrtf(fmaxf(ss[i / D], 1e-12f));
        o[i] = res_s[i] * iv;
    }
}

__global__ __launch_bounds__(256) void vlad_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ feat,
                                                       const float* __restrict__ A, const float* __restrict__ centers,
                                                       const float* __restrict__ out, const float* __restrict__ inv,
                                                       float* __restrict__ dfeat, float* __restrict__ dscore,
                                                       float* __restrict__ dcenters, int L, int D, int K, int KG,
                                                       int accumulate_dfeat) {
    extern __shared__ float sm[];
    const int DP = D + 1;
    float* A_s = sm;                    // [L][KG]
    float* dres = A_s + L * KG;         // [K][D+1]
    float* asum = dres + K * DP;        // [KG]
    float* dot = asum + KG;             // [K]   n.do
    float* cd = dot + K;                // [K]   dres.c
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = feat + (long)b * L * D;
    const float* dO = dout + (long)b * K * D;
    const float* n = out + (long)b * K * D;
    for (int i = tid; i < L * KG; i += 256) A_s[i] = A[(long)b * L * KG + i];
    for (int k = tid; k < K; k += 256) { dot[k] = 0.f; cd[k] = 0.f; }
    __syncthreads();
    for (int k = tid; k < KG; k += 256) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += A_s[l * KG + k];
        asum[k] = s;
    }
    // n . dO per cluster (wave per cluster)
    for (int k = wave; k < K; k += 4) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s = fmaf(n[(long)k * D + d], dO[(long)k * D + d], s);
        s = wave_sum(s);
        if (lane == 0) dot[k] = s;
    }
    __syncthreads();
    // dres = l2-normalise backward; dcenters; dres.c
    for (int k = wave; k < K; k += 4) {
        const float iv = inv[(long)b * K + k];
        const float aiv = fabsf(iv), dt = iv > 0.f ? dot[k] : 0.f;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float g = (dO[(long)k * D + d] - n[(long)k * D + d] * dt) * aiv;
            dres[k * DP + d] = g;
            const float c = centers[(long)k * D + d];
            s = fmaf(g, c, s);
            atomicAdd(dcenters + (long)k * D + d, -asum[k] * g);
        }
        s = wave_sum(s);
        if (lane == 0) cd[k] = s;
    }
    __syncthreads();
    // dA row by row (wave per row), then softmax backward -> dscore.  lane owns clusters lane and lane+64.
    for (int l = wave; l < L; l += 4) {
        float xr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int d = lane + i * 64; xr[i] = d < D ? x[(long)l * D + d] : 0.f; }
        float dA0 = 0.f, dA1 = 0.f;
        for (int k0 = 0; k0 < K; k0 += 16) {                        // 16 independent butterfly chains per batch
            float part[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float s = 0.f;
                if (k0 + j < K) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const int d = lane + i * 64; if (d < D) s = fmaf(dres[(k0 + j) * DP + d], xr[i], s); }
                }
                part[j] = s;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int j = 0; j < 16; ++j) part[j] += __shfl_xor(part[j], off, 64);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (k0 + j < K) {
                    const float s = part[j] - cd[k0 + j];
                    if (k0 + j == lane) dA0 = s;
                    if (k0 + j == lane + 64) dA1 = s;
                }
        }
        const float a0 = lane < KG ? A_s[l * KG + lane] : 0.f;
        const float a1 = lane + 64 < KG ? A_s[l * KG + lane + 64] : 0.f;
        const float t = wave_sum(dA0 * a0 + dA1 * a1);
        if (lane < KG) dscore[((long)b * L + l) * KG + lane] = a0 * (dA0 - t);
        if (lane + 64 < KG) dscore[((long)b * L + l) * KG + lane + 64] = a1 * (dA1 - t);
    }
    // dfeat[l,d] = sum_k A[l,k] dres[k,d]
    for (int d = tid; d < D; d += 256) {
        for (int l = 0; l < L; ++l) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) s = fmaf(A_s[l * KG + k], dres[k * DP + d], s);
            float* dst = dfeat + ((long)b * L + l) * D + d;
            *dst = accumulate_dfeat ? *dst + s : s;
        }
    }
}

// ------------------------------------------------------------------ MFMA formulation (D <= 256, K <= 64)
// The three contractions of the layer are small per-utterance GEMMs -- V = A^T X (K x L x D), dA = X dres^T (L x D x K),
// dfeat = A dres (L x K x D) -- so they run on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains) from LDS tiles instead of scalar
// FMA loops with one LDS read per FMA.  L is processed in chunks (softmax, A_out and the column sums are chunk-local), so
// utterances of any length fit.  Operand conventions of the 32x32x2 MFMA: A-operand lane -> (row lane&31, k lane>>5),
// B-operand lane -> (col lane&31, k lane>>5), acc[r] -> (row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31).
// Stage `nrows` rows of a (rows, D) global matrix into an LDS tile [nrows_tile][ld] (ld >= round_up(D, 32)), zero-filling the
// rows beyond `rows` and the columns beyond D: a wave per row, float4 global loads when the row is 16-byte aligned.
template <int NWV = 4>
__device__ __forceinline__ void stage_rows(float* __restrict__ tile, int ld, const float* __restrict__ src, int D, int rows,
                                           int nrows_tile, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int Dpad = ((D + 31) / 32) * 32;
    const bool vec = (D & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    if (vec && Dpad <= 256) {                                         // one float4 per lane and row: eight rows per trip, loads first
        const int d = lane * 4;                                       // (row by row this was 8-16 dependent L2 round trips per chunk)
        // (round 4: SIXTEEN rows per trip -- a 64-row tile in ONE round trip; the features were written by another XCD's launch, a trip to
        // them is ~4 000 cycles: in-kernel stamps, tools/vlad_timing.py)
        constexpr int RPT = 64 / NWV;                                 // rows per wave and trip
        for (int r0 = wave; r0 < nrows_tile; r0 += 64) {
            float4 v[RPT];
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int r = r0 + NWV * j;
                const bool ok = r < rows && d < D;
                v[j] = ok ? *reinterpret_cast<const float4*>(src + (long)r * D + d) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int r = r0 + NWV * j;
                if (r < nrows_tile && d < Dpad) { float* dst = tile + r * ld + d; dst[0] = v[j].x; dst[1] = v[j].y; dst[2] = v[j].z; dst[3] = v[j].w; }
            }
        }
        return;
    }
    for (int r = wave; r < nrows_tile; r += NWV) {
        float* dst = tile + r * ld;
        if (r < rows && vec) {
            for (int d = lane * 4; d < Dpad; d += 256) {
                const float4 v = d < D ? *reinterpret_cast<const float4*>(src + (long)r * D + d) : make_float4(0.f, 0.f, 0.f, 0.f);
                dst[d] = v.x; dst[d + 1] = v.y; dst[d + 2] = v.z; dst[d + 3] = v.w;      // ld may be odd: scalar LDS stores
            }
        } else {
            for (int d = lane; d < Dpad; d += 64) dst[d] = (r < rows && d < D) ? src[(long)r * D + d] : 0.f;
        }
    }
}

constexpr int VF_LC = 64;           // rows of L per chunk, forward
constexpr int VB_LC = 32;           // backward (two more LDS tiles live)

// Softmax over the KG scores of the 64 rows of a chunk into A_c (zero beyond `rows` / KG) + the staging of the chunk's features (stage_rows),
// whose loads go out between the score loads and their use.  SMX: columns per lane (KG <= 4 * SMX).
template <int SMX, int NWV>
__device__ __forceinline__ void assign_rows(float* __restrict__ A_c, int KGP, const float* __restrict__ sc, int KG, int l0, int rows,
                                            float* __restrict__ X_c, int DP, const float* __restrict__ x, int D, int tid) {
    const int srow_ = (tid & 255) >> 2, sq_ = tid & 3;
    const bool on = tid < 256;                                       // (wave-uniform: the first four waves)
    float sv[SMX];
    float smax = -INFINITY;
    const float* srp = sc + (long)(l0 + min(srow_, rows - 1)) * KG;
#pragma unroll
    for (int j = 0; j < SMX; ++j) {
        const int k = sq_ + 4 * j;
        sv[j] = (on && k < KG && srow_ < rows) ? srp[k] : -INFINITY;
        smax = fmaxf(smax, sv[j]);
    }
    stage_rows<NWV>(X_c, DP, x + (long)l0 * D, D, rows, VF_LC, tid);
    if (!on) return;
    smax = fmaxf(smax, SAR_DPP_F(smax, 0xB1)); smax = fmaxf(smax, SAR_DPP_F(smax, 0x4E));
    float ssum = 0.f;
#pragma unroll
    for (int j = 0; j < SMX; ++j) {
        const float e = srow_ < rows ? __expf(sv[j] - smax) : 0.f;   // exp(-inf) = 0 for the columns past KG
        sv[j] = e; ssum += e;
    }
    ssum += SAR_DPP_F(ssum, 0xB1); ssum += SAR_DPP_F(ssum, 0x4E);
    const float sinv = srow_ < rows ? 1.f / ssum : 0.f;
    for (int k = sq_; k < KGP; k += 4) A_c[srow_ * KGP + k] = 0.f;   // (columns KG .. KGP, and the rows past `rows`)
#pragma unroll
    for (int j = 0; j < SMX; ++j) {
        const int k = sq_ + 4 * j;
        if (k < KG) A_c[srow_ * KGP + k] = sv[j] * sinv;
    }
}

// EIGHT waves, one 32-column tile of D each (round 4; was four waves with two tiles each): the kernel runs ONCE per step on 64 workgroups -- mostly
// straight-line code with one wave per SIMD, ~8 cycles an instruction (tools/vlad_timing.py: 47 000 cycles for 5 900 instructions) -- so what a
// wave does not have to execute is what it saves: half the staging, half the accumulator entries to turn into residuals, to reduce and to store.
constexpr int VF_NT = 512, VF_NW = 8;
__global__ __launch_bounds__(VF_NT) void vlad_fwd_mfma_kernel(const float* __restrict__ feat, const float* __restrict__ score,
                                                            const float* __restrict__ centers, float* __restrict__ out,
                                                            float* __restrict__ A_out, float* __restrict__ inv_out, int L,
                                                            int D, int K, int KG) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Kt = (K + 31) / 32, Dt = (D + 31) / 32, KGP = max(KG, Kt * 32), DP = Dt * 32;
    float* A_c = sm;                     // [VF_LC][KGP]   soft assignments of the chunk (zero beyond L / KG)
    float* X_c = A_c + VF_LC * KGP;      // [VF_LC][DP]    features of the chunk (zero beyond L / D)
    float* asum = X_c + VF_LC * DP;      // [KGP]
    float* ss = asum + KGP;              // [64]           squared norms of the residual rows
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = feat + (long)b * L * D;
    const float* sc = score + (long)b * L * KG;
    float* Ao = A_out + (long)b * L * KG;
    TSTAMP(0); TSTAMP(1);
    for (int i = tid; i < KGP; i += VF_NT) asum[i] = 0.f;
    for (int i = tid; i < 64; i += VF_NT) ss[i] = 0.f;
    f32x16 acc[1][2];                    // [d-tile slot: wave (+ 8: D <= 256 has none)][k-tile]
    float cen[2][16][1];                 // centres at the accumulator entries [k-tile][r][d-tile slot] (constant indices only: registers)
#pragma unroll
    for (int a = 0; a < 1; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][c][i] = 0.f;
    for (int l0 = 0; l0 < L; l0 += VF_LC) {
        const int rows = min(VF_LC, L - l0);
        __syncthreads();                                             // previous chunk's tiles are no longer read
        // Soft assignments: FOUR lanes per row (64 rows x 4 lanes: the first four waves), scores straight from global memory into registers,
        // quad reductions on DPP -- requested in front of the feature staging so that both round trips overlap.  (Round 4: was scores -> LDS,
        // barrier, ONE thread per row walking its columns three times through LDS.)  Instantiated for 36 / 64 / 128 columns: straight-line
        // code costs this kernel ~8 cycles an instruction.
        if (KG <= 36) assign_rows<9, VF_NW>(A_c, KGP, sc, KG, l0, rows, X_c, DP, x, D, tid);
        else if (KG <= 64) assign_rows<16, VF_NW>(A_c, KGP, sc, KG, l0, rows, X_c, DP, x, D, tid);
        else assign_rows<32, VF_NW>(A_c, KGP, sc, KG, l0, rows, X_c, DP, x, D, tid);
        __syncthreads();
        TSTAMP(2);
        for (int i = tid; i < rows * KG; i += VF_NT) { const int r = i / KG, k = i - r * KG; Ao[(long)(l0 + r) * KG + k] = A_c[r * KGP + k]; }
        for (int i = tid; i < 4 * KG; i += VF_NT) {                    // column sums: four row groups per column, LDS atomics (was one thread
            const int kc = i >> 2, rg = i & 3;                       // per column walking all rows)
            float t = 0.f;
            for (int r = rg; r < rows; r += 4) t += A_c[r * KGP + kc];
            atomicAdd(&asum[kc], t);
        }
        const int i31 = lane & 31, kk = lane >> 5;
        TSTAMP(3);
        if (l0 == 0) {                                                // the centres of this lane's accumulator entries: requested under the MFMA loop
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int a = 0; a < 1; ++a) {
                        const int k = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), d = (wave + VF_NW * a) * 32 + (lane & 31);
                        cen[c][r][a] = (c < Kt && (wave + VF_NW * a) < Dt && d < D && k < K) ? centers[(long)k * D + d] : 0.f;
                    }
        }
        for (int r0 = 0; r0 < VF_LC; r0 += 2) {
            if (r0 >= rows) break;                                    // block-uniform; further rows are zero
            float av[2], bv[1];
#pragma unroll
            for (int c = 0; c < 2; ++c) av[c] = c < Kt ? A_c[(r0 + kk) * KGP + c * 32 + i31] : 0.f;
#pragma unroll
            for (int a = 0; a < 1; ++a) bv[a] = (wave + VF_NW * a) < Dt ? X_c[(r0 + kk) * DP + (wave + VF_NW * a) * 32 + i31] : 0.f;
#pragma unroll
            for (int a = 0; a < 1; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if ((wave + VF_NW * a) < Dt && c < Kt) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[a], acc[a][c], 0, 0, 0);
        }
    }
    TSTAMP(4);
    __syncthreads();                                                 // asum complete
    // residuals in registers, squared row norms through one LDS atomic per (half-wave, row)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (c >= Kt) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float sq = 0.f;
#pragma unroll
            for (int a = 0; a < 1; ++a) {
                const int d = (wave + VF_NW * a) * 32 + (lane & 31);
                float v = 0.f;
                if ((wave + VF_NW * a) < Dt && d < D && k < K) v = acc[a][c][r] - asum[k] * cen[c][r][a];
                acc[a][c][r] = v; sq = fmaf(v, v, sq);
            }
            // sum over the 32 lanes of a half-wave: four DPP steps give every lane its 16-lane row's sum, one cross-row exchange the rest
            // (five ds_bpermute round trips per value before, 160 per thread)
            sq += SAR_DPP_F(sq, 0xB1); sq += SAR_DPP_F(sq, 0x4E); sq += SAR_DPP_F(sq, 0x141); sq += SAR_DPP_F(sq, 0x140);
            if ((lane & 15) == 0 && k < K) atomicAdd(&ss[k], sq);    // one LDS atomic per 16-lane row (the cross-row exchange was a ds_bpermute round trip per value)
        }
    }
    TSTAMP(5);
    __syncthreads();
    TSTAMP(6);
    for (int k = tid; k < K; k += VF_NT) {
        const float s2 = ss[k];
        const float iv = 1.f / sqrtf(fmaxf(s2, 1e-12f));
        inv_out[(long)b * K + k] = s2 >= 1e-12f ? iv : -iv;
    }
    float* o = out + (long)b * K * D;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (c >= Kt) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (k >= K) continue;
            const float iv = 1.f / sqrtf(fmaxf(ss[k], 1e-12f));
#pragma unroll
            for (int a = 0; a < 1; ++a) {
                const int d = (wave + VF_NW * a) * 32 + (lane & 31);
                if ((wave + VF_NW * a) < Dt && d < D) o[(long)k * D + d] = acc[a][c][r] * iv;
            }
        }
    }
    TSTAMP(7); TSTAMP(15);
}

// (EIGHT waves like the forward kernel, round 4: every per-wave share -- clusters of the normalisation backward, d-steps of dA, the d-tile of
// dfeat -- halves, and the softmax backward runs on sixteen lanes per row instead of one.)
constexpr int VB_NT = 512, VB_NW = 8;
__global__ __launch_bounds__(VB_NT) void vlad_bwd_mfma_kernel(const float* __restrict__ dout, const float* __restrict__ feat,
                                                            const float* __restrict__ A, const float* __restrict__ centers,
                                                            const float* __restrict__ out, const float* __restrict__ inv,
                                                            float* __restrict__ dfeat, float* __restrict__ dscore,
                                                            float* __restrict__ dcenters, int L, int D, int K, int KG,
                                                            int accumulate_dfeat) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Kt = (K + 31) / 32, Dt = (D + 31) / 32, KP = Kt * 32, DP1 = Dt * 32 + 1, KGP1 = max(KG, KP) | 1;
    float* dres = sm;                     // [KP][DP1]       gradient w.r.t. the un-normalised residuals (zero beyond K / D)
    float* X_c = dres + KP * DP1;         // [VB_LC][DP1]
    float* A_c = X_c + VB_LC * DP1;       // [VB_LC][KGP1]
    float* dA_s = A_c + VB_LC * KGP1;     // [VB_LC][KP + 1]
    float* asum = dA_s + VB_LC * (KP + 1);  // [KG]
    float* dot = asum + KG;               // [K]   n . dO
    float* cd = dot + K;                  // [K]   dres . c
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = feat + (long)b * L * D;
    const float* Ab = A + (long)b * L * KG;
    const float* dO = dout + (long)b * K * D;
    const float* n = out + (long)b * K * D;
    for (int i = tid; i < KP * DP1; i += VB_NT) dres[i] = 0.f;
    // The prologue loops are chains of memory round trips (a workgroup per utterance: nothing else hides them), so they are
    // unrolled to keep independent loads in flight (as plain loops: 82 us for the kernel, unrolled 71 us)
    // One workgroup per (utterance, chunk of VB_LC frames): everything behind dres -- dA, the softmax backward, dfeat -- is private to a
    // frame, and the centre gradient is linear in the column sums of A, so each chunk adds the share of ITS frames.  (One workgroup per
    // utterance looped over the chunks: 32 workgroups x 6 chunks in cfg5, 258 us; the dres prologue is repeated per chunk instead.)
    const int l_first = blockIdx.y * VB_LC, l_end = min(L, l_first + VB_LC);
    // the centre gradient -asum[k] * dres[k][d] is the same dres for every chunk of the utterance: chunk 0 adds it once with the column
    // sums over ALL frames (128 workgroups adding into the same 8 192 addresses were half of this kernel: device-scope atomics on one
    // address serialise)
    const bool centres = blockIdx.y == 0;
    for (int k = tid; k < KG; k += VB_NT) {                            // column sums of the saved assignments (chunk 0 only)
        float t = 0.f;
        if (centres) {
#pragma unroll 13
            for (int l = 0; l < L; ++l) t += Ab[(long)l * KG + k];
        }
        asum[k] = t;
    }
    __syncthreads();                                                 // asum complete, dres zeroed
    // l2-normalise backward; dcenters; dres . c -- a wave takes its clusters four at a time with every load of the four (n, dO, centres,
    // inv: 13 per lane and cluster) in flight at once, and keeps n . dO in registers (one pass, no barrier in between).  Cluster by
    // cluster this was 2 x K/4 dependent round trips in front of the MFMA work: ~40 of the kernel's 48 us.
    for (int k0 = wave; k0 < K; k0 += 4 * VB_NW) {
        float nv[4][4], dv[4][4], cv[4][4], ivv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = k0 + VB_NW * j;
            const bool kv = kk < K;
            ivv[j] = kv ? inv[(long)b * K + kk] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = lane + 64 * i;
                const bool ok = kv && d < D;
                const long o = ok ? (long)kk * D + d : 0;
                nv[j][i] = ok ? n[o] : 0.f; dv[j][i] = ok ? dO[o] : 0.f; cv[j][i] = ok ? centers[o] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = k0 + VB_NW * j;
            if (kk >= K) break;                                      // wave-uniform
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) t = fmaf(nv[j][i], dv[j][i], t);
            t = wave_sum(t);
            const float aiv = fabsf(ivv[j]), dt = ivv[j] > 0.f ? t : 0.f, as = asum[kk];
            float u = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = lane + 64 * i;
                if (d < D) {
                    const float g = (dv[j][i] - nv[j][i] * dt) * aiv;
                    dres[kk * DP1 + d] = g;
                    u = fmaf(g, cv[j][i], u);
                    if (centres) atomicAdd(dcenters + (long)kk * D + d, -as * g);
                }
            }
            u = wave_sum(u);
            if (lane == 0) cd[kk] = u;
        }
    }
    const int i31 = lane & 31, kk = lane >> 5;
    {
        const int l0 = l_first, rows = l_end - l_first;
        __syncthreads();                                             // dres complete / previous chunk consumed
        stage_rows<VB_NW>(X_c, DP1, x + (long)l0 * D, D, rows, VB_LC, tid);
        for (int i = tid; i < VB_LC * KGP1; i += VB_NT) {
            const int r = i / KGP1, k = i - r * KGP1;
            A_c[i] = (r < rows && k < KG) ? Ab[(long)(l0 + r) * KG + k] : 0.f;
        }
        for (int i = tid; i < VB_LC * (KP + 1); i += VB_NT) dA_s[i] = 0.f;
        __syncthreads();
        // dA[l][k] = sum_d X[l][d] dres[k][d]: each wave reduces a quarter of d, partial tiles meet in LDS
        {
            f32x16 pa[2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) pa[c][i] = 0.f;
            const int dsteps = (Dt * 32) / 2, per = (dsteps + VB_NW - 1) / VB_NW;
            for (int st = wave * per; st < min(dsteps, (wave + 1) * per); ++st) {
                const int d = st * 2 + kk;
                const float av = X_c[i31 * DP1 + d];
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (c < Kt) pa[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, dres[(c * 32 + i31) * DP1 + d], pa[c], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (c < Kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        atomicAdd(&dA_s[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * (KP + 1) + c * 32 + (lane & 31)], pa[c][r]);
        }
        __syncthreads();
        {                                                            // softmax backward -> dscore: SIXTEEN lanes per row (32 rows x 16 = the block)
            const int r = tid >> 4, q = tid & 15;
            const float* arow = A_c + r * KGP1;
            const float* drow = dA_s + r * (KP + 1);
            float t = 0.f;
            for (int k2 = q; k2 < K; k2 += 16) t = fmaf(arow[k2], drow[k2] - cd[k2], t);
            t += SAR_DPP_F(t, 0xB1); t += SAR_DPP_F(t, 0x4E); t += SAR_DPP_F(t, 0x141); t += SAR_DPP_F(t, 0x140);   // every lane: its 16-lane row's sum
            if (r < rows) {
                float* ds = dscore + ((long)b * L + l0 + r) * KG;
                for (int k2 = q; k2 < KG; k2 += 16) ds[k2] = arow[k2] * ((k2 < K ? drow[k2] - cd[k2] : 0.f) - t);
            }
        }
        // dfeat[l][d] = sum_k A[l][k] dres[k][d]: d-tile `wave`; reduction over the K (padded) clusters
        {
            f32x16 fa[1];
#pragma unroll
            for (int a = 0; a < 1; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) fa[a][i] = 0.f;
            for (int k0 = 0; k0 < KP; k0 += 2) {
                const float av = (k0 + kk) < K ? A_c[i31 * KGP1 + k0 + kk] : 0.f;
#pragma unroll
                for (int a = 0; a < 1; ++a)
                    if ((wave + VB_NW * a) < Dt) fa[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, dres[(k0 + kk) * DP1 + (wave + VB_NW * a) * 32 + i31], fa[a], 0, 0, 0);
            }
#pragma unroll
            for (int a = 0; a < 1; ++a)
                if ((wave + VB_NW * a) < Dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int l = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), d = (wave + VB_NW * a) * 32 + (lane & 31);
                        if (l < rows && d < D) {
                            float* dst = dfeat + ((long)b * L + l0 + l) * D + d;
                            *dst = accumulate_dfeat ? *dst + fa[a][r] : fa[a][r];
                        }
                    }
        }
    }
}

}  // namespace

extern "C" int sar_vlad_fwd_f32(const float* feat, const float* score, const float* centers, float* out, float* A,
                                float* inv, int B, int L, int D, int K, int G, hipStream_t stream) {
    const int KG = K + G;
    if (!feat || !score || !centers || !out || !A || !inv || K < 1 || G < 0) return SAR_EINVAL;
    if (D <= 256 && K <= 64 && KG <= 128) {             // MFMA formulation
        const int Kt = (K + 31) / 32, Dt = (D + 31) / 32, KGP = KG > Kt * 32 ? KG : Kt * 32;
        const size_t ldsm = sizeof(float) * ((size_t)VF_LC * KGP + (size_t)VF_LC * Dt * 32 + KGP + 64);
        static PerDeviceFlag attr_set;  // raise the dynamic-LDS cap once (host-side attribute, not a stream op)
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(vlad_fwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SAR_ELAUNCH;
            attr_set = true;
        }
        hipLaunchKernelGGL(vlad_fwd_mfma_kernel, dim3(B), dim3(VF_NT), ldsm, stream, feat, score, centers, out, A, inv, L, D, K, KG);
        SAR_CHECK_LAUNCH();
        return SAR_OK;
    }
    const size_t lds = sizeof(float) * ((size_t)L * KG + KG + K + (size_t)K * D);
    if (lds > 160 * 1024) return SAR_EINVAL;
    static size_t lds_set = 0;     // raise the dynamic-LDS cap once per size (host-side attribute, not a stream op)
    if (lds > 48 * 1024 && lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(vlad_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SAR_ELAUNCH;
        lds_set = 160 * 1024;
    }
    hipLaunchKernelGGL(vlad_fwd_kernel, dim3(B), dim3(256), lds, stream, feat, score, centers, out, A, inv, L, D, K, KG);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_vlad_bwd_f32(const float* dout, const float* feat, const float* A, const float* centers,
                                const float* out, const float* inv, float* dfeat, float* dscore, float* dcenters, int B,
                                int L, int D, int K, int G, int accumulate_dfeat, hipStream_t stream) {
    const int KG = K + G;
    if (!dout || !feat || !A || !centers || !out || !inv || !dfeat || !dscore || !dcenters || KG > 128 || D > 512) return SAR_EINVAL;
    if (D <= 256 && K <= 64) {                          // MFMA formulation
        const int Kt = (K + 31) / 32, Dt = (D + 31) / 32, KP = Kt * 32, DP1 = Dt * 32 + 1, KGP1 = (KG > KP ? KG : KP) | 1;
        const size_t ldsm = sizeof(float) * ((size_t)KP * DP1 + (size_t)VB_LC * DP1 + (size_t)VB_LC * KGP1 + (size_t)VB_LC * (KP + 1) + KG + 2 * K);
        static PerDeviceFlag attr_set;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(vlad_bwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SAR_ELAUNCH;
            attr_set = true;
        }
        hipLaunchKernelGGL(vlad_bwd_mfma_kernel, dim3(B, (L + VB_LC - 1) / VB_LC), dim3(VB_NT), ldsm, stream, dout, feat, A, centers, out, inv, dfeat, dscore, dcenters, L, D, K, KG, accumulate_dfeat);
        SAR_CHECK_LAUNCH();
        return SAR_OK;
    }
    const size_t lds = sizeof(float) * ((size_t)L * KG + (size_t)K * (D + 1) + KG + 2 * K);
    if (lds > 160 * 1024) return SAR_EINVAL;
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(vlad_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SAR_ELAUNCH;
        lds_set = 160 * 1024;
    }
    hipLaunchKernelGGL(vlad_bwd_kernel, dim3(B), dim3(256), lds, stream, dout, feat, A, centers, out, inv, dfeat, dscore, dcenters, L, D, K, KG, accumulate_dfeat);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

#ifdef SAR_TIMING
extern "C" int sar_dbg_set_timing_buffer_vlad(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tbuf_vlad), &buf, sizeof(buf)) == hipSuccess ? SAR_OK : SAR_ELAUNCH;
}
#endif
