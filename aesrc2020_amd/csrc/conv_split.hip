// Split-precision convolution on PRE-SPLIT operands (gfx950).
//
// conv_igemm.hip's bf16x3 kernels convert fp32 -> (hi, lo) bf16 pairs while staging tiles into LDS; in-kernel timing
// (tools/conv_timing.py) shows their k-loop is bound by exactly that VALU work (~125 instructions per thread per k-tile
// for 6 MFMAs), and every element is converted again by every block that touches it (9 taps x Cout/64 column tiles).
// Here each operand is split ONCE where it is produced:
//   * activations:  relu(bn(x)) -> planes xh, xl (bf16, NHWC like x)      sar_bn_apply_split_bf16   (one pass, shared by the
//                                                                          forward conv and the weight gradient)
//   * gradients:    dy          -> planes dyh, dyl                          written by the BN-backward apply kernel
//   * weights:      w (HWIO)    -> wt_{h,l} [Cout][K] (k contiguous: forward B operand)
//                                  wd_{h,l} [K][Cout] (native order: dgrad B operand)        sar_conv_weight_split_bf16
// and the conv kernels stage tiles with plain 16-byte copies (no arithmetic), three v_mfma_f32_32x32x16_bf16 per product
// block (lo*hi + hi*lo + hi*hi), fp32 accumulate, fp32 outputs.  Same numerics as conv_igemm.hip's bf16x3 path.
// Reference ops: Conv2D at resnet.py:39-42,60-63,82-87,113-117 with the BN+ReLU in front of it (resnet.py:22-26).
#include "common.h"

namespace {

#ifdef SAR_TIMING   // in-kernel phase timing, tools/conv_timing.py only (see conv_igemm.hip)
__device__ unsigned long long* g_tbuf_s = nullptr;
#define TSTAMP(slot)                                                                                                   \
    do {                                                                                                               \
        if (g_tbuf_s && threadIdx.x == 0) {                                                                            \
            const long bl = (long)blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);          \
            g_tbuf_s[bl * 16 + (slot)] = ((slot) == 0 || (slot) == 15) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
        }                                                                                                              \
    } while (0)
#else
#define TSTAMP(slot) do {} while (0)
#endif

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ unsigned fdiv(unsigned m, unsigned d, unsigned magic) { return d == 1 ? m : __umulhi(m, magic); }
__device__ __forceinline__ unsigned outer_mask(unsigned hm, unsigned wm, int KH, int KW) {
    unsigned v = 0;
    for (int kh = 0; kh < KH; ++kh)
        if ((hm >> kh) & 1u) v |= wm << (kh * KW);
    return v;
}
__device__ __forceinline__ uint4 sel4(bool ok, uint4 v) {
    v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
    return v;
}

// ------------------------------------------------------------------ operand producers
// xh/xl = split([relu](x * scale[c] + shift[c])); 8 consecutive channels per thread (C % 8 == 0).  scale == nullptr: identity.
__global__ __launch_bounds__(256) void bn_apply_split_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, __bf16* __restrict__ xh,
                                                             __bf16* __restrict__ xl, long n8, int C, int relu) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const long base = i * 8;
        const int c = (int)(base % C);
        float v[8];
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(x + base);
        *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(x + base + 4);
        if (scale) {
            float sc[8], sh[8];
            *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(scale + c);
            *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(scale + c + 4);
            *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(shift + c);
            *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(shift + c + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
        }
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = relu ? fmaxf(v[j], 0.f) : v[j];
            const __bf16 h = (__bf16)f;
            hi[j] = h; lo[j] = (__bf16)(f - (float)h);
        }
        *reinterpret_cast<bf16x8*>(xh + base) = hi; *reinterpret_cast<bf16x8*>(xl + base) = lo;
    }
}

// One launch splits the kernels of all convolutions of the trunk: table row = {w, wt_h, wt_l, wd_h, wd_l, K, Cout, first block}.
struct WSplitDesc { const float* w; __bf16* wt_h; __bf16* wt_l; __bf16* wd_h; __bf16* wd_l; long K, Cout, blk0; };

__global__ __launch_bounds__(256) void weight_split_kernel(const WSplitDesc* __restrict__ table, int n) {
    __shared__ __bf16 th[64][34], tl[64][34];                         // [n][k] transposed tile (+2 pad)
    int d = 0;
    while (d + 1 < n && (long)blockIdx.x >= table[d + 1].blk0) ++d;
    const WSplitDesc ds = table[d];
    const int K = (int)ds.K, Cout = (int)ds.Cout;
    const int tiles_n = Cout / 64;
    const int t = blockIdx.x - (int)ds.blk0, k0 = (t / tiles_n) * 32, n0 = (t % tiles_n) * 64;
    const int tid = threadIdx.x;
    // read 32 (k) x 64 (n) fp32, coalesced along n: thread -> k = tid / 16 (+16), n = (tid % 16) * 4
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = tid / 16 + h * 16, nn = (tid % 16) * 4;
        const float4 v = *reinterpret_cast<const float4*>(ds.w + (long)(k0 + k) * Cout + n0 + nn);
        const float f[4] = {v.x, v.y, v.z, v.w};
        __bf16 hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { hi[j] = (__bf16)f[j]; lo[j] = (__bf16)(f[j] - (float)hi[j]); th[nn + j][k] = hi[j]; tl[nn + j][k] = lo[j]; }
        __bf16* oh = ds.wd_h + (long)(k0 + k) * Cout + n0 + nn; __bf16* ol = ds.wd_l + (long)(k0 + k) * Cout + n0 + nn;
#pragma unroll
        for (int j = 0; j < 4; ++j) { oh[j] = hi[j]; ol[j] = lo[j]; }
    }
    __syncthreads();
    // write wt[n][k0 .. k0+31]: thread -> n = tid / 4, 8 consecutive k
    const int nn = tid / 4, kk = (tid % 4) * 8;
    bf16x8 vh, vl;
#pragma unroll
    for (int j = 0; j < 8; ++j) { vh[j] = th[nn][kk + j]; vl[j] = tl[nn][kk + j]; }
    *reinterpret_cast<bf16x8*>(ds.wt_h + (long)(n0 + nn) * K + k0 + kk) = vh;
    *reinterpret_cast<bf16x8*>(ds.wt_l + (long)(n0 + nn) * K + k0 + kk) = vl;
}

// ------------------------------------------------------------------ forward
struct FwdSP {
    const __bf16* xh; const __bf16* xl; const __bf16* wth; const __bf16* wtl;
    const float* bias; const float* res; float* y;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pt, pl, splitk;
    unsigned magic_a, magic_b;
};

// Block tile (64*TM) x 64 x 32, 4 waves as 2 x 2, TM 32x32 accumulators per wave.  LDS tiles [row][k] bf16 with 80-byte rows.
template <int TM>
__global__ __launch_bounds__(256) void conv_fwd_split_kernel(FwdSP p) {
    constexpr int BMt = 64 * TM, BNt = 64, BK = 32, BKP = BK + 8;
    __shared__ __attribute__((aligned(16))) __bf16 Ah[2][BMt][BKP];
    __shared__ __attribute__((aligned(16))) __bf16 Al[2][BMt][BKP];
    __shared__ __attribute__((aligned(16))) __bf16 Bh[2][BNt][BKP];
    __shared__ __attribute__((aligned(16))) __bf16 Bl[2][BNt][BKP];
    TSTAMP(0); TSTAMP(1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned M = (unsigned)(p.B * p.Ho * p.Wo);
    const int K = p.KH * p.KW * p.Cin;
    const unsigned m0 = blockIdx.x * BMt;
    const int n0 = blockIdx.y * BNt;
    const int ch = (tid & 3) * 8;                                     // 16-byte chunk (8 bf16) inside the 32-wide k-tile
    int abase[TM]; unsigned avalid[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const unsigned m = m0 + (tid >> 2) + i * 64;
        abase[i] = 0; avalid[i] = 0;
        if (m < M) {
            const unsigned t = fdiv(m, p.Wo, p.magic_a), wo = m - t * (unsigned)p.Wo;
            const unsigned n = fdiv(t, p.Ho, p.magic_b), ho = t - n * (unsigned)p.Ho;
            const int h0 = (int)ho * p.stride - p.pt, w0 = (int)wo * p.stride - p.pl;
            abase[i] = (((int)n * p.H + h0) * p.W + w0) * p.Cin + ch;
            unsigned hm = 0, wmk = 0;
            for (int kk = 0; kk < p.KH; ++kk) hm |= (unsigned)(h0 + kk >= 0 && h0 + kk < p.H) << kk;
            for (int kk = 0; kk < p.KW; ++kk) wmk |= (unsigned)(w0 + kk >= 0 && w0 + kk < p.W) << kk;
            avalid[i] = outer_mask(hm, wmk, p.KH, p.KW);
        }
    }
    const int b_n = tid >> 2;
    const bool bcol_ok = n0 + b_n < p.Cout;
    const long b_off = (long)min(n0 + b_n, p.Cout - 1) * K + ch;     // clamped: always a valid address
    struct Regs { uint4 ah[TM], al[TM], bh, bl; unsigned vmask; };
    Regs R0, R1; R0.vmask = R1.vmask = 0;
    auto load = [&](Regs& R, int kt) {
        const int k0 = kt * BK;
        const int tap = k0 / p.Cin, ci0 = k0 - tap * p.Cin;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const int tapoff = (kh * p.W + kw) * p.Cin + ci0;
        R.vmask = 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned ok = (avalid[i] >> tap) & 1u;
            const int off = ok ? abase[i] + tapoff : 0;
            R.ah[i] = *reinterpret_cast<const uint4*>(p.xh + off);
            R.al[i] = *reinterpret_cast<const uint4*>(p.xl + off);
            R.vmask |= ok << i;
        }
        R.bh = *reinterpret_cast<const uint4*>(p.wth + b_off + k0);
        R.bl = *reinterpret_cast<const uint4*>(p.wtl + b_off + k0);
    };
    auto store = [&](const Regs& R, int buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const bool ok = (R.vmask >> i) & 1u;                      // padding taps are exactly zero
            *reinterpret_cast<uint4*>(&Ah[buf][(tid >> 2) + i * 64][ch]) = sel4(ok, R.ah[i]);
            *reinterpret_cast<uint4*>(&Al[buf][(tid >> 2) + i * 64][ch]) = sel4(ok, R.al[i]);
        }
        *reinterpret_cast<uint4*>(&Bh[buf][b_n][ch]) = sel4(bcol_ok, R.bh);
        *reinterpret_cast<uint4*>(&Bl[buf][b_n][ch]) = sel4(bcol_ok, R.bl);
    };
    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tm][i] = 0.f;
    const int nkt_all = K / BK, per = (nkt_all + p.splitk - 1) / p.splitk;
    const int kt0 = blockIdx.z * per, nkt = min(nkt_all, kt0 + per);
    const int kg = (lane >> 5) * 8, rc = lane & 31;
    auto mma = [&](int cur) {
#pragma unroll
        for (int st = 0; st < BK / 16; ++st) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bh[cur][wn * 32 + rc][st * 16 + kg]);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bl[cur][wn * 32 + rc][st * 16 + kg]);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Ah[cur][(wm * TM + tm) * 32 + rc][st * 16 + kg]);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(&Al[cur][(wm * TM + tm) * 32 + rc][st * 16 + kg]);
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[tm], 0, 0, 0);      // small terms first
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[tm], 0, 0, 0);
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[tm], 0, 0, 0);
            }
        }
    };
    TSTAMP(2);
    if (kt0 < nkt) {
        const int last = nkt - 1;
        load(R0, kt0); load(R1, min(kt0 + 1, last));
        TSTAMP(3);
        store(R0, 0); load(R0, min(kt0 + 2, last));
        __syncthreads();
        TSTAMP(4);
        for (int kt = kt0;;) {                                           // two explicit exits: see conv_igemm.hip
            mma(0); store(R1, 1); load(R1, min(kt + 3, last));
            __syncthreads();
            if (kt + 1 >= nkt) break;
            mma(1); store(R0, 0); load(R0, min(kt + 4, last));
            __syncthreads();
            kt += 2;
            if (kt >= nkt) break;
        }
    }
    TSTAMP(5);
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= p.Cout) return;
    const bool lead = blockIdx.z == 0;
    const float bv = (p.bias && lead) ? p.bias[col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned row = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row >= M) continue;
            const long o = (long)row * p.Cout + col;
            float v = acc[tm][r] + bv;
            if (p.res && lead) v += p.res[o];
            if (p.splitk > 1) atomicAdd(p.y + o, v);
            else p.y[o] = v;
        }
#ifdef SAR_TIMING
    __builtin_amdgcn_s_waitcnt(0);
#endif
    TSTAMP(6); TSTAMP(15);
}

static int pick_splitk(int tiles, int nkt) {
    if (tiles >= 200) return 1;
    int s = (768 + tiles - 1) / tiles;
    if (s > nkt / 4) s = nkt / 4;
    if (s > 4) s = 4;
    return s < 1 ? 1 : s;
}

}  // namespace

#ifdef SAR_TIMING
extern "C" int sar_dbg_set_timing_buffer_split(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tbuf_s), &buf, sizeof(buf)) == hipSuccess ? SAR_OK : SAR_ELAUNCH;
}
#endif

extern "C" int sar_bn_apply_split_bf16(const float* x, const float* scale, const float* shift, void* xh, void* xl, long n,
                                       int C, int relu, hipStream_t stream) {
    if (!x || !xh || !xl || (scale && !shift) || C % 8 || n % 8 || n < 0) return SAR_EINVAL;
    if (n == 0) return SAR_OK;
    const long n8 = n / 8;
    const int nb = (int)(n8 / 256 + 1 < 4096 ? n8 / 256 + 1 : 4096);
    hipLaunchKernelGGL(bn_apply_split_kernel, dim3(nb), dim3(256), 0, stream, x, scale, shift, static_cast<__bf16*>(xh),
                       static_cast<__bf16*>(xl), n8, C, relu);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_conv_weight_split_bf16(const void* table, int n, int total_blocks, hipStream_t stream) {
    if (!table || n < 1 || total_blocks < 1) return SAR_EINVAL;
    hipLaunchKernelGGL(weight_split_kernel, dim3(total_blocks), dim3(256), 0, stream, static_cast<const WSplitDesc*>(table), n);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_conv2d_fwd_split_f32(const void* xh, const void* xl, const void* wth, const void* wtl, const float* bias,
                                        const float* residual, float* y, int B, int H, int W, int Cin, int Ho, int Wo,
                                        int Cout, int KH, int KW, int stride, int pad_top, int pad_left, hipStream_t stream) {
    const long in_e = (long)B * H * W * Cin, out_e = (long)B * Ho * Wo * Cout;
    if (!xh || !xl || !wth || !wtl || !y || Cin % 32 || Cout % 16 || KH * KW > 16 || in_e >= (1L << 31) || out_e >= (1L << 31) ||
        (long)B * Ho * Wo * (Wo > Ho ? Wo : Ho) >= (1L << 32)) return SAR_EINVAL;
    FwdSP p{static_cast<const __bf16*>(xh), static_cast<const __bf16*>(xl), static_cast<const __bf16*>(wth),
            static_cast<const __bf16*>(wtl), bias, residual, y, B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_top, pad_left, 1,
            (unsigned)(((1UL << 32) + Wo - 1) / Wo), (unsigned)(((1UL << 32) + Ho - 1) / Ho)};
    const long M = (long)B * Ho * Wo;
    if (cdiv(M, 128) * cdiv(Cout, 64) >= 300) {
        hipLaunchKernelGGL((conv_fwd_split_kernel<2>), dim3(cdiv(M, 128), cdiv(Cout, 64), 1), dim3(256), 0, stream, p);
    } else {
        const int tiles = cdiv(M, 64) * cdiv(Cout, 64);
        p.splitk = pick_splitk(tiles, KH * KW * Cin / 32);
        if (p.splitk > 1 && sar_zero_async(y, sizeof(float) * M * Cout, stream) != hipSuccess) return SAR_ELAUNCH;
        hipLaunchKernelGGL((conv_fwd_split_kernel<1>), dim3(cdiv(M, 64), cdiv(Cout, 64), p.splitk), dim3(256), 0, stream, p);
    }
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
