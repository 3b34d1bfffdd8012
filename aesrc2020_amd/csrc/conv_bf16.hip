// bf16 ResNet trunk of SAR-Net on gfx950 (BASELINE configs[1] "bf16"): 2-byte NHWC activations / activation gradients in
// HBM, fp32 master weights, fp32 statistics and accumulation, single-pass v_mfma_f32_32x32x16_bf16.
// Reference: resnet.py:22-125 (_bn_relu / _bn_relu_conv / basic_block / _shortcut), TF-SAME padding per SURVEY App. A.
//
// Design (what the round-1 fp32-storage kernels taught): at B=64 a conv launch is ~1 us of matrix time
// per CU, so everything that is not `ds_read + MFMA` has to leave the k-loop.
//  * conv (forward AND input-gradient, one kernel): a block owns TM consecutive output pixels x TN channels.  The padded
//    input patch those pixels touch (rows x padded width x ALL input channels, halo = explicit zeros) is brought into LDS
//    ONCE; BatchNorm+ReLU of the producer (forward) or the BatchNorm backward of the consumer (gradient) is evaluated in
//    that fill -- once per patch element instead of once per tap.  The 9 taps are then nine constant LDS offsets: the k-loop is
//    ds_read_b128 (A fragment) + one 1-KiB coalesced global load of a pre-packed weight fragment (B) + MFMA.  No im2col
//    address arithmetic, no masks, no conversions in the loop.
//  * BatchNorm statistics travel as fp32 atomics into NB=16 bins per layer (conv epilogue -> bins -> every consumer block
//    sums the 16 bins itself in its prologue): no finalize / combine / apply launches.
//  * stride-2 input gradients run as stride-1 convolutions over the zero-dilated gradient (3 + 3 launches per step).
//  * weight gradient: block = 9 waves (one per tap) x one (32 input, 32 output channel) pair x a slice of the pixels; the
//    pixel-major LDS images of activation patch and output gradient feed the MFMA through ds_read_b64_tr_b16 (hardware
//    transpose), so neither operand is transposed in registers; partial tiles are added with fp32 atomics.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int NB = 16;          // statistic bins per BatchNorm (fp32 atomics; contention = blocks / 16 per address)
constexpr int CMAX = 512;       // max channels of a BatchNorm / conv side
constexpr int CST_BYTES = 6 * CMAX * 4;

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

// ---------------------------------------------------------------------------------------------------- geometry
// Output pixel m = (b, ho, wo) reads the padded virtual input at (b*Hp + s*ho + dy, s*wo + dx), dy, dx < K.  The virtual
// input is the source tensor dilated by d (zeros between samples; d = 2 only for stride-2 input gradients) and padded by
// (pt, pl) zeros in front.
struct Geo {
    int B, Hs, Ws, Cin;         // source tensor [B][Hs][Ws][Cin] (bf16)
    int Ho, Wo;                 // output pixels per image
    int K, s, d, pt, pl;
    int Hp, Wp;                 // padded extent: s*(Ho-1)+K, s*(Wo-1)+K
    int M;                      // B*Ho*Wo
    unsigned mg_wo, mg_howo, mg_wp, mg_hp;
};
// n / d, branch-free: mg = floor(2^32/d) (2^32-1 for d = 1) underestimates the quotient by at most 1
__device__ __forceinline__ unsigned udiv(unsigned n, unsigned mg, unsigned d) {
    const unsigned q = __umulhi(n, mg);
    return q + ((q + 1) * d <= n ? 1u : 0u);
}
__device__ __forceinline__ void pix_rp(const Geo& g, unsigned m, unsigned& Rp, unsigned& wq) {
    const unsigned hw = (unsigned)(g.Ho * g.Wo);
    const unsigned b = udiv(m, g.mg_howo, hw), r = m - b * hw;
    const unsigned ho = udiv(r, g.mg_wo, (unsigned)g.Wo), wo = r - ho * g.Wo;
    Rp = b * g.Hp + g.s * ho; wq = g.s * wo;
}
// element offset of padded position (Rp, wp) in the source tensor, or NONE: padding / dilation hole / past the batch / !in
constexpr unsigned NONE = 0xffffffffu;
__device__ __forceinline__ unsigned src_off(const Geo& g, unsigned Rp, unsigned wp, bool in) {
    const unsigned b = udiv(Rp, g.mg_hp, (unsigned)g.Hp), hp = Rp - b * g.Hp;
    const int u = (int)hp - g.pt, v = (int)wp - g.pl, dm = g.d - 1;          // d in {1, 2}: mask and shift are d-1
    const int us = u >> dm, vs = v >> dm;
    const bool ok = in & ((u | v) >= 0) & (((u | v) & dm) == 0) & (b < (unsigned)g.B) & (us < g.Hs) & (vs < g.Ws);
    const unsigned o = ((b * g.Hs + us) * g.Ws + vs) * g.Cin;                 // < 2^31 elements (launcher checks)
    return ok ? o : NONE;
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

// cst[0..5][CMAX]: scale, shift, mean, rstd, c2 = S1/N, c3 = S2/N for channels [c0, c0+nc).  mode 1: forward constants
// (from the bins, or from `saved` when there are none: inference, or a backward consumer); mode 2: backward constants.
__device__ __forceinline__ void bn_prologue(const BnP& bn, int mode, int C, int c0, int nc, float* cst, bool owner, int nthreads) {
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
        cst[i] = scale; cst[CMAX + i] = shift; cst[2 * CMAX + i] = mean; cst[3 * CMAX + i] = rstd;
        if (mode == 2) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) { s1 += (double)bn.bbins[(b * 2) * C + c]; s2 += (double)bn.bbins[(b * 2 + 1) * C + c]; }
            cst[4 * CMAX + i] = (float)(s1 * (double)bn.inv_n); cst[5 * CMAX + i] = (float)(s2 * (double)bn.inv_n);
            if (owner && bn.write && bn.dgamma != nullptr) { bn.dgamma[c] += (float)s2; bn.dbeta[c] += (float)s1; }
        }
    }
}

// Per-thread constants of 8 consecutive channels (a 16-byte piece), read once from the LDS table.
struct Cst8 { float sc[8], sh[8], mu[8], rs[8], c2[8], c3[8]; };
__device__ __forceinline__ void load_cst8(const float* cst, int ch, int mode, Cst8& k) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        k.sc[j] = cst[ch + j]; k.sh[j] = cst[CMAX + ch + j];
        if (mode == 2) { k.mu[j] = cst[2 * CMAX + ch + j]; k.rs[j] = cst[3 * CMAX + ch + j]; k.c2[j] = cst[4 * CMAX + ch + j]; k.c3[j] = cst[5 * CMAX + ch + j]; }
    }
}
// MODE 0: copy; 1: relu(x*scale+shift); 2: BatchNorm(+ReLU) backward  dx = scale*(dz - c2 - xhat*c3), dz = da*[z>0]
template <int MODE>
__device__ __forceinline__ uint4 xform(const uint4 q, const uint4 q2, const Cst8& k) {
    if constexpr (MODE == 0) return q;
    float a[8], o[8];
    unpack8(q, a);
    if constexpr (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaxf(fmaf(a[j], k.sc[j], k.sh[j]), 0.f);
    } else {
        float y[8];
        unpack8(q2, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = fmaf(y[j], k.sc[j], k.sh[j]);
            const float dz = z > 0.f ? a[j] : 0.f;
            const float xh = (y[j] - k.mu[j]) * k.rs[j];
            o[j] = k.sc[j] * (dz - k.c2[j] - xh * k.c3[j]);
        }
    }
    return pack8(o);
}

// ---------------------------------------------------------------------------------------------------- conv (fwd / dgrad)
struct ConvP {
    Geo g; int Cout;
    const u16* x; const u16* x2; const bf16x8* wpk; const float* bias; const u16* res; u16* y;
    int fill; BnP fb;
    int epi; float* e_bins; const u16* e_y; const float* e_saved;
    int lg_pc;                  // log2(Cin/8): 16-byte pieces per position
    int rs;                     // LDS row stride in bytes = Cin*2 + 16 (conflict-free ds_read_b128 across rows)
    int Fs;                     // fragment stride of the weight stream per 32 output channels: K*K*Cin/16 + 8 (zero tail)
    int cap_rows;               // patch rows the LDS allocation holds
};

template <int MODE>
__device__ __forceinline__ void fill_patch(const Geo& g, const u16* __restrict__ x, const u16* __restrict__ x2, int lg_pc, int rs,
                                           unsigned char* patch, const float* cst, unsigned Rp_lo, unsigned nrows, int tid) {
    const unsigned pc = 1u << lg_pc, total = (nrows * g.Wp) << lg_pc;
    const unsigned dummy = nrows * g.Wp * rs + (tid & 63) * 16;   // 1 KiB behind the patch (the launcher allocates it)
    const unsigned c8 = tid & (pc - 1);                            // 256 % pc == 0: a thread keeps its channel group
    Cst8 k;
    if constexpr (MODE != 0) load_cst8(cst, c8 * 8, MODE, k);
    for (unsigned i0 = tid; i0 < total; i0 += 4 * 256) {
        uint4 q[4], q2[4]; unsigned off[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned i = i0 + 256 * j, pos = i >> lg_pc;
            const unsigned prow = udiv(pos, g.mg_wp, (unsigned)g.Wp), wp = pos - prow * g.Wp;
            off[j] = src_off(g, Rp_lo + prow, wp, i < total);
            lo[j] = pos * rs + c8 * 16;
            const unsigned o = off[j] == NONE ? 0u : off[j] + c8 * 8;
            q[j] = *reinterpret_cast<const uint4*>(x + o);
            if constexpr (MODE == 2) q2[j] = *reinterpret_cast<const uint4*>(x2 + o); else q2[j] = q[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 r = xform<MODE>(q[j], q2[j], k);                 // branch-free: pieces past the end land in a scratch slot
            *reinterpret_cast<uint4*>(patch + (i0 + 256 * j < total ? lo[j] : dummy)) = off[j] == NONE ? make_uint4(0, 0, 0, 0) : r;
        }
    }
}

template <int NKC>
__device__ __forceinline__ void load_afrags(bf16x8 (&a)[NKC], const unsigned char* patch, unsigned addr) {
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) a[kc] = *reinterpret_cast<const bf16x8*>(patch + addr + kc * 32);
}
// One (virtual) tap: NKC MFMAs on the fragments in `a`; bpt = this tap's first weight fragment; the weight ring bq
// (depth 8, this tap starts at slot s0 -- a constant after unrolling) is refilled 8 fragments ahead.
template <int NKC>
__device__ __forceinline__ void tap_mma(f32x16& acc, const bf16x8 (&a)[NKC], bf16x8 (&bq)[8], const bf16x8* bpt, const int s0) {
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
        const int slot = (s0 + kc) & 7;
        const bf16x8 Bv = bq[slot];
        bq[slot] = bpt[(size_t)(kc + 8) * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kc], Bv, acc, 0, 0, 0);
    }
}

// WN: waves along the output channels (tile 32*(4/WN) pixels x 32*WN channels); NKC: 16-channel blocks per virtual tap;
// VT: virtual taps = K*K x channel halves (a tap of Cin <= 256 channels is one virtual tap, Cin = 512 is two): 1, 2, 9, 18.
template <int WN, int NKC, int VT>
__global__ __launch_bounds__(256) void conv_bf16_kernel(const ConvP p) {
    constexpr int WM = 4 / WN, TM = 32 * WM, TN = 32 * WN;
    constexpr int CP = TN / 2, RG = 256 / CP, RPT = TM / RG;       // epilogue: column pairs, row groups, rows per thread
    constexpr int KS = VT % 9 == 0 ? 3 : 1, NH = VT / (KS * KS);  // taps per side, channel halves
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* cst = reinterpret_cast<float*>(smem);
    unsigned char* patch = smem + CST_BYTES;
    const Geo& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w / WN, wn = w % WN;
    const unsigned m0 = blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    const unsigned M = (unsigned)g.M;

    if (p.fill) {
        bn_prologue(p.fb, p.fill, g.Cin, 0, g.Cin, cst, blockIdx.x == 0 && blockIdx.y == 0, 256);
        __syncthreads();
    }
    unsigned Rp_lo, Rp_hi, wtmp;
    pix_rp(g, m0, Rp_lo, wtmp);
    pix_rp(g, min(m0 + TM - 1, M - 1), Rp_hi, wtmp);
    Rp_hi += g.K - 1;
    unsigned nrows = Rp_hi - Rp_lo + 1;
    if (nrows > (unsigned)p.cap_rows) nrows = p.cap_rows;          // never taken: the launcher sizes cap_rows as an upper bound

    // ---- weight stream of this wave's 32 output channels: first 8 fragments in flight before the fill
    const bf16x8* bp = p.wpk + ((size_t)(blockIdx.y * WN + wn) * p.Fs) * 64 + lane;
    bf16x8 bq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bq[j] = bp[(size_t)j * 64];

    // ---- fill: the padded patch, all input channels, transformed once
    if (p.fill == 0) fill_patch<0>(g, p.x, p.x2, p.lg_pc, p.rs, patch, cst, Rp_lo, nrows, tid);
    else if (p.fill == 1) fill_patch<1>(g, p.x, p.x2, p.lg_pc, p.rs, patch, cst, Rp_lo, nrows, tid);
    else fill_patch<2>(g, p.x, p.x2, p.lg_pc, p.rs, patch, cst, Rp_lo, nrows, tid);
    // ---- per-lane A address
    unsigned a_lane;
    {
        unsigned Rp, wq;
        pix_rp(g, min(m0 + wm * 32 + (lane & 31), M - 1), Rp, wq);
        unsigned prow = Rp - Rp_lo;
        if (prow + g.K > nrows) prow = 0;                          // only with the clamp above
        a_lane = (prow * g.Wp + wq) * p.rs + (lane >> 5) * 16;
    }
    __syncthreads();

    // ---- k-phase, straight-line (no loop-carried register rings for the compiler to copy around): virtual taps in stream
    // order; the A fragments of the NEXT virtual tap are read from LDS while the MFMAs of the current one issue (two register
    // sets), weight fragments arrive 8 MFMAs ahead.
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 afr[2][NKC];
    load_afrags<NKC>(afr[0], patch, a_lane);
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        if (v + 1 < VT) {
            const int tap = (v + 1) / NH, hh = (v + 1) % NH, dy = tap / KS, dx = tap % KS;
            load_afrags<NKC>(afr[(v + 1) & 1], patch, a_lane + (unsigned)((dy * g.Wp + dx) * p.rs + hh * (NKC * 32)));
        }
        tap_mma<NKC>(acc, afr[v & 1], bq, bp + (size_t)(v * NKC) * 64, (v * NKC) & 7);
    }
    __syncthreads();                                               // the patch is dead: reuse it as the output staging tile

    // ---- epilogue: accumulators -> LDS (fp32) -> row-major pass: + bias (+ residual), bf16 stores, statistics
    float* ot = reinterpret_cast<float*>(patch);                   // [TM][TN]
    float* part = ot + TM * TN;                                    // [RG][TN][2]
    {
        const int col = wn * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            ot[row * TN + col] = acc[r];
        }
    }
    __syncthreads();
    const int cp = tid % CP, rg = tid / CP;
    const int c = n0 + 2 * cp;
    const float b0 = p.bias ? p.bias[c] : 0.f, b1 = p.bias ? p.bias[c + 1] : 0.f;
    float sc0 = 0.f, sc1 = 0.f, sh0 = 0.f, sh1 = 0.f, mu0 = 0.f, mu1 = 0.f, rs0 = 0.f, rs1 = 0.f;
    if (p.epi == 2) {
        const float* sv = p.e_saved; const int C = p.Cout;
        mu0 = sv[c]; mu1 = sv[c + 1]; rs0 = sv[C + c]; rs1 = sv[C + c + 1];
        sc0 = sv[2 * C + c]; sc1 = sv[2 * C + c + 1]; sh0 = sv[3 * C + c]; sh1 = sv[3 * C + c + 1];
    }
    unsigned rv[RPT], yv[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const unsigned m = min(m0 + rg + RG * i, M - 1);
        const size_t o = (size_t)m * p.Cout + c;
        rv[i] = p.res ? *reinterpret_cast<const unsigned*>(p.res + o) : 0u;
        yv[i] = p.epi == 2 ? *reinterpret_cast<const unsigned*>(p.e_y + o) : 0u;
    }
    float s10 = 0.f, s11 = 0.f, s20 = 0.f, s21 = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int row = rg + RG * i;
        const unsigned m = m0 + row;
        const float2 a = *reinterpret_cast<const float2*>(ot + row * TN + 2 * cp);
        float v0 = a.x + b0 + bf_lo(rv[i]), v1 = a.y + b1 + bf_hi(rv[i]);
        const unsigned pk = pack2(v0, v1);
        if (m < M) {
            *reinterpret_cast<unsigned*>(p.y + (size_t)m * p.Cout + c) = pk;
            v0 = bf_lo(pk); v1 = bf_hi(pk);                        // statistics of what was stored
            if (p.epi == 1) { s10 += v0; s11 += v1; s20 += v0 * v0; s21 += v1 * v1; }
            if (p.epi == 2) {
                const float y0 = bf_lo(yv[i]), y1 = bf_hi(yv[i]);
                const float d0 = fmaf(y0, sc0, sh0) > 0.f ? v0 : 0.f, d1 = fmaf(y1, sc1, sh1) > 0.f ? v1 : 0.f;
                s10 += d0; s11 += d1; s20 += d0 * (y0 - mu0) * rs0; s21 += d1 * (y1 - mu1) * rs1;
            }
        }
    }
    if (p.epi == 0) return;                                        // kernel-uniform
    part[(rg * TN + 2 * cp) * 2 + 0] = s10; part[(rg * TN + 2 * cp) * 2 + 1] = s20;
    part[(rg * TN + 2 * cp + 1) * 2 + 0] = s11; part[(rg * TN + 2 * cp + 1) * 2 + 1] = s21;
    __syncthreads();
    if (tid < 2 * TN) {
        const int cc = tid >> 1, which = tid & 1;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < RG; ++r) s += part[(r * TN + cc) * 2 + which];
        atomicAdd(p.e_bins + ((blockIdx.x % NB) * 2 + which) * p.Cout + n0 + cc, s);
    }
}

// ---------------------------------------------------------------------------------------------------- weight gradient
struct WgP {
    Geo g; int Cout;
    const u16* x; int a_fill; BnP ab;
    const u16* dy; const u16* dy2; int d_fill; BnP db;
    float* dw; float* dbias;
    int nchunks, cpb;           // 64-pixel chunks in total / per block
    int cap_pos;                // patch positions one LDS buffer holds
};
constexpr int WG_MAXA = 4;      // 16-byte patch pieces per thread and chunk (9-wave blocks)

__device__ __forceinline__ s16x4 lds_tr(const unsigned char* base, unsigned off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
}

// KK = taps = waves per block; AF / DF = fill modes of the activation patch (0, 1) and of dY (0, 2).
// grid (pixel slices, Cin/32, Cout/32).
template <int KK, int AF, int DF>
__global__ __launch_bounds__(KK * 64) void wgrad_bf16_kernel(const WgP p) {
    constexpr int NT = KK * 64;
    constexpr int MAXA = KK == 1 ? 4 : WG_MAXA;
    constexpr int ND = (256 + NT - 1) / NT;                        // dY pieces (64 pixels x 4) per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* cstA = reinterpret_cast<float*>(smem);                  // [6][CMAX] layout, 32 channels used
    float* cstD = cstA + 6 * CMAX;
    unsigned char* bufs = smem + 2 * CST_BYTES;
    const Geo& g = p.g;
    const unsigned buf_bytes = 256 + 64 * 64 + (unsigned)p.cap_pos * 64 + 1024;   // qtab | dY [64][32] | patch [pos][32] | scratch
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
    const unsigned M = (unsigned)g.M;
    const int c_first = blockIdx.x * p.cpb, c_end = min(c_first + p.cpb, p.nchunks);

    if constexpr (AF != 0) bn_prologue(p.ab, 1, g.Cin, ci0, 32, cstA, false, NT);
    if constexpr (DF != 0) bn_prologue(p.db, 2, p.Cout, co0, 32, cstD, blockIdx.x == 0 && blockIdx.y == 0, NT);
    if constexpr (AF != 0 || DF != 0) __syncthreads();
    const int c8 = tid & 3;                                        // NT % 4 == 0: fixed channel group per thread
    Cst8 kA, kD;
    if constexpr (AF != 0) load_cst8(cstA, c8 * 8, 1, kA);
    if constexpr (DF != 0) load_cst8(cstD, c8 * 8, 2, kD);

    uint4 qa[MAXA]; unsigned offa[MAXA];
    uint4 qd[ND], qd2[ND];
    unsigned Rp_lo = 0, npos = 0;
    const u16* __restrict__ xs = p.x + ci0 + c8 * 8;
    const u16* __restrict__ dys = p.dy + co0 + c8 * 8;
    const u16* __restrict__ dy2s = p.dy2 + co0 + c8 * 8;

    auto issue = [&](int ch) {
        const unsigned mc0 = (unsigned)ch * 64;
        if constexpr (KK == 1) {                                   // 1x1: the "patch" is the 64 pixels themselves
            npos = 64;
#pragma unroll
            for (int j = 0; j < MAXA; ++j) {
                const unsigned pos = (tid + NT * j) >> 2;
                unsigned Rp, wq;
                pix_rp(g, min(mc0 + pos, M - 1), Rp, wq);
                offa[j] = src_off(g, Rp, wq, mc0 + pos < M);
                qa[j] = *reinterpret_cast<const uint4*>(xs + (offa[j] == NONE ? 0u : offa[j]));
            }
        } else {
            unsigned Rp_hi, wtmp;
            pix_rp(g, mc0, Rp_lo, wtmp);
            pix_rp(g, min(mc0 + 63, M - 1), Rp_hi, wtmp);
            npos = min((Rp_hi + g.K - Rp_lo) * g.Wp, (unsigned)p.cap_pos);
#pragma unroll
            for (int j = 0; j < MAXA; ++j) {
                const unsigned pos = (tid + NT * j) >> 2;
                const unsigned prow = udiv(pos, g.mg_wp, (unsigned)g.Wp), wp = pos - prow * g.Wp;
                offa[j] = src_off(g, Rp_lo + prow, wp, pos < npos);
                qa[j] = *reinterpret_cast<const uint4*>(xs + (offa[j] == NONE ? 0u : offa[j]));
            }
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const unsigned m = mc0 + ((tid + NT * j) >> 2);
            const size_t o = (size_t)min(m, M - 1) * p.Cout;
            qd[j] = *reinterpret_cast<const uint4*>(dys + o);
            if constexpr (DF != 0) qd2[j] = *reinterpret_cast<const uint4*>(dy2s + o); else qd2[j] = qd[j];
        }
    };
    auto store = [&](int ch, unsigned char* buf) {
        const unsigned mc0 = (unsigned)ch * 64;
        unsigned* qtab = reinterpret_cast<unsigned*>(buf);
        unsigned char* dyt = buf + 256;
        unsigned char* ap = buf + 256 + 64 * 64;
        unsigned char* dump = ap + (unsigned)p.cap_pos * 64 + (tid & 63) * 16;
#pragma unroll
        for (int j = 0; j < MAXA; ++j) {
            const unsigned pos = (tid + NT * j) >> 2;
            const uint4 r = xform<AF>(qa[j], qa[j], kA);                // branch-free: unused pieces land in a scratch slot
            *reinterpret_cast<uint4*>(pos < npos ? ap + pos * 64 + c8 * 16 : dump) = offa[j] == NONE ? make_uint4(0, 0, 0, 0) : r;
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const unsigned i = tid + NT * j, m = mc0 + (i >> 2);
            const uint4 r = xform<DF>(qd[j], qd2[j], kD);
            *reinterpret_cast<uint4*>(i < 256 ? dyt + (i >> 2) * 64 + c8 * 16 : dump) = m < M ? r : make_uint4(0, 0, 0, 0);
        }
        if (tid < 64) {
            unsigned q = tid * 64;
            if constexpr (KK != 1) {
                unsigned Rp, wq;
                pix_rp(g, min(mc0 + tid, M - 1), Rp, wq);
                q = ((Rp - Rp_lo) * g.Wp + wq) * 64;
                if (mc0 + tid >= M || q + ((g.K - 1) * g.Wp + g.K) * 64 > npos * 64) q = 0;   // past the batch (dY row is zero) / cap_pos clamp
            }
            qtab[tid] = q;
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = p.dbias != nullptr && w == 0 && blockIdx.y == 0;
    const unsigned toff = KK == 1 ? 0u : (unsigned)(((w / g.K) * g.Wp + (w % g.K)) * 64);
    const int gi = lane & 15, gg = lane >> 4, cb = gg & 1, kh = gg >> 1;
    const unsigned colb = (16 * cb + 4 * (gi & 3)) * 2;
    const unsigned prb = 8 * kh + (gi >> 2);

    if (c_first < c_end) issue(c_first);
    for (int ch = c_first; ch < c_end; ++ch) {
        unsigned char* buf = bufs + ((ch - c_first) & 1) * buf_bytes;
        store(ch, buf);
        __syncthreads();
        if (ch + 1 < c_end) issue(ch + 1);
        const unsigned* qtab = reinterpret_cast<const unsigned*>(buf);
        const unsigned char* dyt = buf + 256;
        const unsigned char* ap = buf + 256 + 64 * 64;
        unsigned q0[4], q1[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { q0[ks] = qtab[16 * ks + prb]; q1[ks] = qtab[16 * ks + prb + 4]; }
        bf16x8 Af[4], Bf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned pr0 = 16 * ks + prb, pr1 = pr0 + 4;
            const s16x4 b_lo = lds_tr(dyt, pr0 * 64 + colb), b_hi = lds_tr(dyt, pr1 * 64 + colb);
            const s16x4 a_lo = lds_tr(ap, q0[ks] + toff + colb), a_hi = lds_tr(ap, q1[ks] + toff + colb);
            Af[ks] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7));
            Bf[ks] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b_lo, b_hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[ks], Bf[ks], acc, 0, 0, 0);
        if (do_bias) {                                             // wave-uniform (scalar branch)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += (float)Bf[ks][j];
        }
    }
    // ---- flush: dw[tap][ci0+row][co0+col] += acc
    {
        const int col = lane & 31;
        float* o = p.dw + ((size_t)w * g.Cin + ci0) * p.Cout + co0 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            atomicAdd(o + (size_t)row * p.Cout, acc[r]);
        }
        if (do_bias) {
            bsum += __shfl_xor(bsum, 32, 64);
            if (lane < 32) atomicAdd(p.dbias + co0 + lane, bsum);
        }
    }
}

// ---------------------------------------------------------------------------------------------------- BatchNorm passes
// Residual-stream gradient: out = BN(+ReLU) backward of `dact` through the BatchNorm whose input was x (+ add).
template <typename DT>
__global__ __launch_bounds__(256) void bn_bwd_apply_bf16_kernel(const DT* __restrict__ dact, const u16* __restrict__ x, const BnP bn,
                                                                const u16* __restrict__ add, u16* __restrict__ out, long R, int C, int relu) {
    __shared__ float cst[6 * CMAX];
    bn_prologue(bn, 2, C, 0, C, cst, blockIdx.x == 0, 256);
    __syncthreads();
    const int pc = C >> 3;                                         // power of two
    const long total = R * pc, stride = (long)gridDim.x * 256;     // stride % pc == 0
    const int c8 = threadIdx.x & (pc - 1);
    Cst8 k;
    load_cst8(cst, c8 * 8, 2, k);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long e = i * 8;
        float a[8], y[8], o[8], ad[8];
        if constexpr (sizeof(DT) == 2) unpack8(*reinterpret_cast<const uint4*>(dact + e), a);
        else {
            const float4 f0 = *reinterpret_cast<const float4*>(dact + e), f1 = *reinterpret_cast<const float4*>(dact + e + 4);
            a[0] = f0.x; a[1] = f0.y; a[2] = f0.z; a[3] = f0.w; a[4] = f1.x; a[5] = f1.y; a[6] = f1.z; a[7] = f1.w;
        }
        unpack8(*reinterpret_cast<const uint4*>(x + e), y);
        if (add) unpack8(*reinterpret_cast<const uint4*>(add + e), ad);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = fmaf(y[j], k.sc[j], k.sh[j]);
            const float dz = (!relu || z > 0.f) ? a[j] : 0.f;
            const float xh = (y[j] - k.mu[j]) * k.rs[j];
            o[j] = k.sc[j] * (dz - k.c2[j] - xh * k.c3[j]) + (add ? ad[j] : 0.f);
        }
        *reinterpret_cast<uint4*>(out + e) = pack8(o);
    }
}

// The two sums of a BatchNorm(+ReLU) backward whose incoming gradient does not come out of a conv epilogue (the final BN).
template <typename DT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_bf16_kernel(const DT* __restrict__ dact, const u16* __restrict__ x, const float* __restrict__ saved,
                                                                 float* __restrict__ bbins, long R, int C, int relu) {
    __shared__ float red[256 * 16];
    const int pc = C >> 3;
    const long total = R * pc, stride = (long)gridDim.x * 256;
    const int c8 = threadIdx.x & (pc - 1);
    float sc[8], sh[8], mu[8], rs[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j;
        mu[j] = saved[c]; rs[j] = saved[C + c]; sc[j] = saved[2 * C + c]; sh[j] = saved[3 * C + c]; s1[j] = 0.f; s2[j] = 0.f;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long e = i * 8;
        float a[8], y[8];
        if constexpr (sizeof(DT) == 2) unpack8(*reinterpret_cast<const uint4*>(dact + e), a);
        else {
            const float4 f0 = *reinterpret_cast<const float4*>(dact + e), f1 = *reinterpret_cast<const float4*>(dact + e + 4);
            a[0] = f0.x; a[1] = f0.y; a[2] = f0.z; a[3] = f0.w; a[4] = f1.x; a[5] = f1.y; a[6] = f1.z; a[7] = f1.w;
        }
        unpack8(*reinterpret_cast<const uint4*>(x + e), y);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = fmaf(y[j], sc[j], sh[j]);
            const float dz = (!relu || z > 0.f) ? a[j] : 0.f;
            s1[j] += dz; s2[j] += dz * (y[j] - mu[j]) * rs[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x * 16 + j] = s1[j]; red[threadIdx.x * 16 + 8 + j] = s2[j]; }
    __syncthreads();
    // threads with equal (tid % pc) hold the same channels: 2*C outputs, each summed over 256/pc threads
    for (int o = threadIdx.x; o < 2 * C; o += 256) {
        const int which = o / C, c = o % C, g8 = c >> 3, j = c & 7;
        float s = 0.f;
        for (int t = g8; t < 256; t += pc) s += red[t * 16 + which * 8 + j];
        atomicAdd(bbins + ((blockIdx.x % NB) * 2 + which) * C + c, s);
    }
}

// Final BatchNorm + ReLU of the trunk (resnet.py:178-181): bf16 in, fp32 out for the sequence layers.
__global__ __launch_bounds__(256) void bn_apply_bf16_f32_kernel(const u16* __restrict__ x, const BnP bn, float* __restrict__ out, long R, int C, int relu) {
    __shared__ float cst[6 * CMAX];
    bn_prologue(bn, 1, C, 0, C, cst, blockIdx.x == 0, 256);
    __syncthreads();
    const int pc = C >> 3;
    const long total = R * pc, stride = (long)gridDim.x * 256;
    const int c8 = threadIdx.x & (pc - 1);
    Cst8 k;
    load_cst8(cst, c8 * 8, 1, k);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long e = i * 8;
        float y[8];
        unpack8(*reinterpret_cast<const uint4*>(x + e), y);
#pragma unroll
        for (int j = 0; j < 8; ++j) { y[j] = fmaf(y[j], k.sc[j], k.sh[j]); if (relu) y[j] = fmaxf(y[j], 0.f); }
        *reinterpret_cast<float4*>(out + e) = make_float4(y[0], y[1], y[2], y[3]);
        *reinterpret_cast<float4*>(out + e + 4) = make_float4(y[4], y[5], y[6], y[7]);
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, u16* __restrict__ y, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const float4 a = *reinterpret_cast<const float4*>(x + i * 8), b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
        *reinterpret_cast<uint4*>(y + i * 8) = make_uint4(pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w));
    }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const u16* __restrict__ x, float* __restrict__ y, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + i * 8), f);
        *reinterpret_cast<float4*>(y + i * 8) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(y + i * 8 + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
}

// ---------------------------------------------------------------------------------------------------- weight packing
// One launch packs the kernels of many convolutions (fp32 HWIO) into the bf16 fragment streams the conv kernel reads:
// stream[nt][f][lane][8], nt = output-channel block of 32, f = (tap, channel block of 16) in consumption order (+ 8 zero
// fragments: the conv kernel prefetches 8 ahead without a bound check), lane l <-> output channel nt*32 + l%32, reduction channels 16*kc + 8*(l/32) + j.
// forward: output = co, reduction = ci, tap as stored;  dgrad: output = ci, reduction = co, taps flipped.
struct PackRow { const float* w; u16* wf; u16* wd; long taps, Cin, Cout, first_block, pad_; };

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackRow* __restrict__ tab, int n) {
    int r = 0;
    while (r + 1 < n && (long)blockIdx.x >= tab[r + 1].first_block) ++r;
    const PackRow row = tab[r];
    const int taps = (int)row.taps, Cin = (int)row.Cin, Cout = (int)row.Cout;
    const int Ff = taps * Cin / 16, Ffp = Ff + 8, Fd = taps * Cout / 16, Fdp = Fd + 8;
    const long nf = (long)(Cout / 32) * Ffp * 64, nd = (long)(Cin / 32) * Fdp * 64;    // 16-byte pieces
    long i = ((long)blockIdx.x - row.first_block) * 256 + threadIdx.x;
    if (i >= nf + nd) return;
    const bool fwd = i < nf;
    if (!fwd) i -= nf;
    const int Fp = fwd ? Ffp : Fdp, F = fwd ? Ff : Fd, Cred = fwd ? Cin : Cout;
    const int lane = (int)(i & 63);
    const long t = i >> 6;
    const int f = (int)(t % Fp), nt = (int)(t / Fp);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (f < F) {
        const int nk = Cred / 16, tap = f / nk, kc = f % nk;
        const int oc = nt * 32 + (lane & 31), rc = kc * 16 + 8 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = fwd ? row.w[((long)tap * Cin + rc + j) * Cout + oc] : row.w[((long)(taps - 1 - tap) * Cin + oc) * Cout + rc + j];
    }
    *reinterpret_cast<uint4*>((fwd ? row.wf : row.wd) + i * 8) = pack8(v);
}

// ---------------------------------------------------------------------------------------------------- host side
static inline unsigned magic(int d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ULL / (unsigned long long)d); }
static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static bool make_geo(Geo& g, int B, int Hs, int Ws, int Cin, int Ho, int Wo, int K, int s, int d, int pt, int pl) {
    if (B < 1 || Hs < 1 || Ws < 1 || Ho < 1 || Wo < 1 || (K != 1 && K != 3) || (s != 1 && s != 2) || (d != 1 && d != 2) || (s == 2 && d == 2)) return false;
    if (pt < 0 || pl < 0) return false;
    g.B = B; g.Hs = Hs; g.Ws = Ws; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.K = K; g.s = s; g.d = d; g.pt = pt; g.pl = pl;
    g.Hp = s * (Ho - 1) + K; g.Wp = s * (Wo - 1) + K;
    const long M = (long)B * Ho * Wo, RP = (long)B * g.Hp;
    if (M >= (1L << 31) || RP * g.Wp >= (1L << 31)) return false;
    g.M = (int)M;
    g.mg_wo = magic(Wo); g.mg_howo = magic(Ho * Wo); g.mg_wp = magic(g.Wp); g.mg_hp = magic(g.Hp);
    return true;
}
// upper bound of the padded rows a run of `tm` consecutive output pixels touches
static int patch_rows_bound(const Geo& g, int tm) {
    const int drg = (tm - 1) / g.Wo + 1;                            // max difference of (image-flattened) output row indices
    const int per_img = g.Hp - g.s * g.Ho > 0 ? g.Hp - g.s * g.Ho : 0;
    return (drg / g.Ho + 1) * per_img + g.s * drg + g.K;
}

template <typename KT>
static bool allow_lds(KT kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

}  // namespace

extern "C" {

// mirrors BnP (include/sarnet.h: sarBnRef)
typedef BnP sarBnRef;

int sar_conv_pack_weights_bf16(const void* table, int n, int total_blocks, hipStream_t stream) {
    if (table == nullptr || n < 1 || total_blocks < 1) return SAR_EINVAL;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(total_blocks), dim3(256), 0, stream, static_cast<const PackRow*>(table), n);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// Generic bf16 convolution: y[M][Cout] = sum over taps, channels of T(src)[padded position] * wpk (+ bias) (+ res), with
//   fill 0: T = identity; 1: T = relu(bn(.)) of `fbn` (forward statistics from fbn->bins, or fbn->saved when bins is null);
//        2: T = BatchNorm(+ReLU) backward of (da = src, y = src2) through `fbn`.
//   epi  0: none; 1: sum / sum of squares of y into e_bins; 2: BatchNorm-backward sums of y (= da) against (e_y, e_saved).
// Forward: d = 1, s = stride, pads as TF.  Input gradient: src = dy, d = stride, s = 1, pt = K-1-pad_top, wpk = dgrad stream.
int sar_conv2d_bf16(const void* src, const void* src2, const void* wpk, const float* bias, const void* res, void* y, int B,
                    int Hs, int Ws, int Cin, int Ho, int Wo, int Cout, int K, int s, int d, int pt, int pl, int fill,
                    const sarBnRef* fbn, int epi, float* e_bins, const void* e_y, const float* e_saved, hipStream_t stream) {
    ConvP p;
    memset(&p, 0, sizeof(p));
    if (!make_geo(p.g, B, Hs, Ws, Cin, Ho, Wo, K, s, d, pt, pl)) return SAR_EINVAL;
    if (!pow2(Cin) || !pow2(Cout) || Cin < 32 || Cout < 32 || Cin > CMAX || Cout > CMAX) return SAR_EINVAL;
    if (fill < 0 || fill > 2 || epi < 0 || epi > 2 || (fill && fbn == nullptr) || (fill == 2 && src2 == nullptr)) return SAR_EINVAL;
    if (epi && e_bins == nullptr) return SAR_EINVAL;
    if (epi == 2 && (e_y == nullptr || e_saved == nullptr)) return SAR_EINVAL;
    if ((long)B * Hs * Ws * Cin >= (1L << 31) || (long)p.g.M * Cout >= (1L << 31)) return SAR_EINVAL;
    p.Cout = Cout;
    p.x = static_cast<const u16*>(src); p.x2 = static_cast<const u16*>(src2); p.wpk = static_cast<const bf16x8*>(wpk);
    p.bias = bias; p.res = static_cast<const u16*>(res); p.y = static_cast<u16*>(y);
    p.fill = fill; if (fbn) p.fb = *fbn;
    p.epi = epi; p.e_bins = e_bins; p.e_y = static_cast<const u16*>(e_y); p.e_saved = e_saved;
    p.lg_pc = ilog2(Cin / 8); p.rs = Cin * 2 + 16;
    p.Fs = K * K * Cin / 16 + 8;
    const int WN = Cout >= 64 ? 2 : 1, TM = 32 * (4 / WN), TN = 32 * WN;
    p.cap_rows = patch_rows_bound(p.g, TM < p.g.M ? TM : p.g.M);
    const size_t patch_bytes = (size_t)p.cap_rows * p.g.Wp * p.rs + 1024, stage_bytes = (size_t)TM * TN * 4 + 256 / (TN / 2) * TN * 2 * 4;
    const size_t lds = CST_BYTES + (patch_bytes > stage_bytes ? patch_bytes : stage_bytes);
    if (lds > 160 * 1024) return SAR_EINVAL;
    const dim3 grid((unsigned)cdiv(p.g.M, TM), (unsigned)(Cout / TN));
    // instantiations: [WN][log2(NKC)-1][K == 3]; Cin = 512 runs NKC = 16 with two channel halves per tap
    typedef void (*kern_t)(const ConvP);
#define CV(WN_, NKC_) {conv_bf16_kernel<WN_, NKC_, 1>, conv_bf16_kernel<WN_, NKC_, 9>}
    static const kern_t kerns[2][5][2] = {{CV(1, 2), CV(1, 4), CV(1, 8), CV(1, 16), {conv_bf16_kernel<1, 16, 2>, conv_bf16_kernel<1, 16, 18>}},
                                          {CV(2, 2), CV(2, 4), CV(2, 8), CV(2, 16), {conv_bf16_kernel<2, 16, 2>, conv_bf16_kernel<2, 16, 18>}}};
#undef CV
    static bool attr[2][5][2];
    const int ki = ilog2(Cin / 16) - 1, k3 = K == 3 ? 1 : 0;
    const kern_t kern = kerns[WN - 1][ki][k3];
    if (!attr[WN - 1][ki][k3]) { if (!allow_lds(kern, 160 * 1024)) return SAR_ELAUNCH; attr[WN - 1][ki][k3] = true; }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// Kernel gradient dw[K][K][Cin][Cout] += T_a(x)^T (*) T_d(dy) (+ dbias += column sums of T_d(dy), nullable).
//   a_fill 0: x as stored; 1: relu(bn(x)) with abn->saved.   d_fill 0: dy as stored; 2: BatchNorm(+ReLU) backward of
//   (da = dy, y = dy2) through dbn (sums from dbn->bbins).  Geometry as the forward conv (s = stride, pads as TF).
int sar_conv2d_wgrad_bf16(const void* x, int a_fill, const sarBnRef* abn, const void* dy, const void* dy2, int d_fill,
                          const sarBnRef* dbn, float* dw, float* dbias, int B, int H, int W, int Cin, int Ho, int Wo, int Cout,
                          int K, int s, int pt, int pl, hipStream_t stream) {
    WgP p;
    memset(&p, 0, sizeof(p));
    if (!make_geo(p.g, B, H, W, Cin, Ho, Wo, K, s, 1, pt, pl)) return SAR_EINVAL;
    if (!pow2(Cin) || !pow2(Cout) || Cin < 32 || Cout < 32 || Cin > CMAX || Cout > CMAX) return SAR_EINVAL;
    if ((a_fill != 0 && a_fill != 1) || (d_fill != 0 && d_fill != 2) || (a_fill && abn == nullptr) || (d_fill && (dbn == nullptr || dy2 == nullptr))) return SAR_EINVAL;
    if ((long)B * H * W * Cin >= (1L << 31) || (long)p.g.M * Cout >= (1L << 31)) return SAR_EINVAL;
    p.Cout = Cout;
    p.x = static_cast<const u16*>(x); p.a_fill = a_fill; if (abn) p.ab = *abn;
    p.dy = static_cast<const u16*>(dy); p.dy2 = static_cast<const u16*>(dy2); p.d_fill = d_fill; if (dbn) p.db = *dbn;
    p.dw = dw; p.dbias = dbias;
    p.nchunks = cdiv(p.g.M, 64);
    const int pairs = (Cin / 32) * (Cout / 32);
    int P = 256 / pairs; if (P < 1) P = 1; if (P > p.nchunks) P = p.nchunks;
    p.cpb = cdiv(p.nchunks, P); P = cdiv(p.nchunks, p.cpb);
    const int NT = K == 3 ? 576 : 64;
    p.cap_pos = K == 3 ? patch_rows_bound(p.g, 64 < p.g.M ? 64 : p.g.M) * p.g.Wp : 64;
    if (K == 3 && p.cap_pos > WG_MAXA * NT / 4) return SAR_EINVAL;    // a chunk's patch must fit the per-thread prefetch registers
    const size_t lds = 2 * CST_BYTES + 2 * (size_t)(256 + 64 * 64 + p.cap_pos * 64 + 1024);
    if (lds > 160 * 1024) return SAR_EINVAL;
    const dim3 grid((unsigned)P, (unsigned)(Cin / 32), (unsigned)(Cout / 32));
    typedef void (*kern_t)(const WgP);
    static const kern_t kerns[2][2][2] = {{{wgrad_bf16_kernel<9, 0, 0>, wgrad_bf16_kernel<9, 0, 2>}, {wgrad_bf16_kernel<9, 1, 0>, wgrad_bf16_kernel<9, 1, 2>}},
                                          {{wgrad_bf16_kernel<1, 0, 0>, wgrad_bf16_kernel<1, 0, 2>}, {wgrad_bf16_kernel<1, 1, 0>, wgrad_bf16_kernel<1, 1, 2>}}};
    static bool attr[2][2][2] = {{{false, false}, {false, false}}, {{false, false}, {false, false}}};
    const int i0 = K == 3 ? 0 : 1, i1 = a_fill ? 1 : 0, i2 = d_fill ? 1 : 0;
    const kern_t kern = kerns[i0][i1][i2];
    if (!attr[i0][i1][i2]) { if (!allow_lds(kern, 160 * 1024)) return SAR_ELAUNCH; attr[i0][i1][i2] = true; }
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, stream, p);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

static int bn_grid(long R, int C) {
    const long pieces = R * (C / 8);
    long nb = (pieces + 256 * 4 - 1) / (256 * 4);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    return (int)nb;
}

int sar_bn_bwd_apply_bf16(const void* dact, int dact_f32, const void* x, const sarBnRef* bn, const void* add, void* out, long R,
                          int C, int relu, hipStream_t stream) {
    if (bn == nullptr || !pow2(C) || C < 8 || C > CMAX || R < 1) return SAR_EINVAL;
    const dim3 grid(bn_grid(R, C));
    if (dact_f32)
        hipLaunchKernelGGL(bn_bwd_apply_bf16_kernel<float>, grid, dim3(256), 0, stream, static_cast<const float*>(dact), static_cast<const u16*>(x), *bn,
                           static_cast<const u16*>(add), static_cast<u16*>(out), R, C, relu);
    else
        hipLaunchKernelGGL(bn_bwd_apply_bf16_kernel<u16>, grid, dim3(256), 0, stream, static_cast<const u16*>(dact), static_cast<const u16*>(x), *bn,
                           static_cast<const u16*>(add), static_cast<u16*>(out), R, C, relu);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

int sar_bn_bwd_reduce_bf16(const void* dact, int dact_f32, const void* x, const float* saved, float* bbins, long R, int C, int relu,
                           hipStream_t stream) {
    if (!pow2(C) || C < 8 || C > CMAX || R < 1) return SAR_EINVAL;
    const dim3 grid(bn_grid(R, C));
    if (dact_f32)
        hipLaunchKernelGGL(bn_bwd_reduce_bf16_kernel<float>, grid, dim3(256), 0, stream, static_cast<const float*>(dact), static_cast<const u16*>(x), saved, bbins, R, C, relu);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_bf16_kernel<u16>, grid, dim3(256), 0, stream, static_cast<const u16*>(dact), static_cast<const u16*>(x), saved, bbins, R, C, relu);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

int sar_bn_apply_bf16_f32(const void* x, const sarBnRef* bn, float* out, long R, int C, int relu, hipStream_t stream) {
    if (bn == nullptr || !pow2(C) || C < 8 || C > CMAX || R < 1) return SAR_EINVAL;
    hipLaunchKernelGGL(bn_apply_bf16_f32_kernel, dim3(bn_grid(R, C)), dim3(256), 0, stream, static_cast<const u16*>(x), *bn, out, R, C, relu);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

int sar_cast_f32_bf16(const float* x, void* y, long n, hipStream_t stream) {
    if (n < 8 || n % 8) return SAR_EINVAL;
    long nb = (n / 8 + 1023) / 1024; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, static_cast<u16*>(y), n / 8);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
int sar_cast_bf16_f32(const void* x, float* y, long n, hipStream_t stream) {
    if (n < 8 || n % 8) return SAR_EINVAL;
    long nb = (n / 8 + 1023) / 1024; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)nb), dim3(256), 0, stream, static_cast<const u16*>(x), y, n / 8);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

int sar_bn_bins(void) { return NB; }

}  // extern "C"
