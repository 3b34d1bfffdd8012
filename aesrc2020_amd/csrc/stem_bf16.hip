// The stem of the bf16 trunk on gfx950: 7x7/2 conv (Cin = 1) -> BatchNorm -> ReLU -> MaxPool 3x3/2 'same'
// (resnet.py:173-174 / 191-192, _conv_bn_relu resnet.py:34-45), forward and backward, WITHOUT the full-resolution map.
//
// At B=64 the conv output is 64 x 200 x 40 x 64 = 131 MB in fp32 (round 1 wrote it once and read it twice: 257 us per step for
// 4 GFLOP).  The conv is 49 multiply-adds per output on an 8 MB input, so it is recomputed wherever it is needed instead:
//   forward  pass 1 (stem16_stats):  conv tile on MFMA -> per-channel sum / sum of squares -> the BatchNorm's statistic bins
//            pass 2 (stem16_pool):   conv tile again -> BatchNorm + ReLU from the bins -> 3x3/2 max-pool from an LDS tile ->
//                                    pooled map (bf16, 16 MB) + argmax slots (u8) -- the only tensors that reach HBM
//   backward pass 1 (stem16_bwd_reduce): the two BatchNorm-backward sums from (gradient, pooled, slots): where ReLU is active the
//                                    pooled value IS gamma*xhat + beta, so xhat follows from it
//            pass 2 (stem16_wgrad):  conv tile a third time -> xhat -> dy = BatchNorm/ReLU/pool backward per pixel (<= 4 candidate
//                                    windows from an LDS image of the pooled window) -> dW += patches^T dy on MFMA, both operands
//                                    through transpose-reads of wave-private pixel-major LDS images; per-block partial tiles,
//                                    summed by stem16_wsum
// MFMA view of the conv (v_mfma_f32_32x32x16_bf16): rows = output channels (A = kernel fragments, held in registers), columns =
// 32 output pixels, k = ky*8 + kx (kx = 7 and ky = 7 are zero kernel entries): the B fragment of a pixel is 8 consecutive bf16 of
// input row 2*oy + ky starting at column 2*ox -- four aligned dwords of the LDS patch, no gather.  Input and kernel are rounded
// to bf16 (the "bf16" mode of BASELINE configs[1]); accumulation, statistics and the kernel gradient are fp32.
#include "trunk16.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

// In-kernel phase timing (tools/stem_timing.py; only with -DSAR_TIMING into tools/libsarnet_timing_stem.so, never part of
// libsarnet_hip.so): thread 0 of every block stamps s_memtime (shader clock) / s_memrealtime (100 MHz, device-wide: slots 0, 15).
#undef TSTAMP                    // (trunk16.h's default; the stem kernels stamp into their own buffer)
#ifdef SAR_TIMING
__device__ unsigned long long* g_tbuf_stem = nullptr;
#define TSTAMP(slot)                                                                                                   \
    do {                                                                                                               \
        if (g_tbuf_stem && threadIdx.x == 0)                                                                           \
            g_tbuf_stem[(long)blockIdx.x * 16 + (slot)] = ((slot) == 0 || (slot) == 15) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define TSTAMP(slot) do {} while (0)
#endif

struct StemG {
    int B, T, D;                // input (B, T, D) fp32 (the reference's (B, T, 80, 1))
    int Ho, Wo, C0;             // conv output; C0 = 32 * NT
    int pt, pl;                 // TF-SAME padding in front
    int Hp, Wp, ppt, ppl;       // pooled output, pool padding in front
    int TR;                     // rows per tile: conv rows (stats, wgrad) / pooled rows (pool)
    int tiles;                  // tiles per image
    int pld;                    // patch row pitch in bf16 elements (even): 2*(Wo-1) + 8
    int patch_bytes;            // LDS bytes reserved for the patch (SAR_X3: three planes -- hi, lo, and the fp16 image the weight gradient multiplies)
    int ppe;                    // SAR_X3: elements (2-byte) from one patch plane to the next
    int fast;                   // D and pl even, pld <= 128: the patch fill may use 8-byte loads, one column pair per thread
    unsigned mg_wo, mg_wp;
};

constexpr int IP = 80;          // bytes per pixel row of a 32-channel bf16 LDS image: 64 + 16, so the 16-byte stores of 16 consecutive pixels hit 64 different banks
constexpr int PLANE = 32 * IP;  // one 32-pixel x 32-channel image
constexpr int ZP = 32 * 4 + 16; // bytes per pixel row of the fp32 activation tile (32 channels per pass): conflict-free 16-byte stores

// sum over the 16 lanes of a DPP row, in every lane: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

// ---- kernel fragments.  wa[ct][s] = w[ky = 2s + h][kx = 0..7][channel ct*32 + lane%32] (zero for ky = 7 / kx = 7), h = lane/32.
// The block gathers them once (7*NT loads per thread, all in flight), rounds to bf16 and parks them in LDS as ready fragments
// [ct*4 + s][lane][16 B]; every wave then reads its NT*4 fragments with ds_read_b128.
template <int NT> struct WStage { float v[NT][7]; };
template <int NT>
__device__ __forceinline__ void stem_w_load(WStage<NT>& ws, const float* __restrict__ w, int tid) {
    constexpr int C0 = NT * 32;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int fl = tid + 256 * i, f = fl >> 6, ln = fl & 63, ct = f >> 2, s = f & 3, ky = min(2 * s + (ln >> 5), 6);
#pragma unroll
        for (int j = 0; j < 7; ++j) ws.v[i][j] = w[(ky * 7 + j) * C0 + ct * 32 + (ln & 31)];
    }
}
template <int NT>
__device__ __forceinline__ void stem_w_store(const WStage<NT>& ws, unsigned char* wfr, int tid) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int fl = tid + 256 * i, f = fl >> 6, ln = fl & 63, s = f & 3;
        const bool ok = 2 * s + (ln >> 5) < 7;
        h16x8 o, ol;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const float v = ok ? ws.v[i][j] : 0.f;
            o[j] = (h16)v;
            if constexpr (X3) ol[j] = (h16)(v - (float)o[j]);
        }
        o[7] = (h16)0.f;
        *reinterpret_cast<h16x8*>(wfr + fl * 16) = o;
        if constexpr (X3) { ol[7] = (h16)0.f; *reinterpret_cast<h16x8*>(wfr + NT * 4096 + fl * 16) = ol; }   // the lo fragments follow the NT*4 hi ones
    }
}
// the (hi) fragments into registers.  (SAR_X3: the lo fragments stay in LDS, behind the hi ones, and stem_mtile reads them where it uses
// them -- held in registers as well they spilled the 64-channel weight-gradient kernel)
template <int NT>
__device__ __forceinline__ void stem_w_frags(h16x8 (&wa)[NT][4], const unsigned char* wfr, int lane) {
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int s = 0; s < 4; ++s) wa[ct][s] = *reinterpret_cast<const h16x8*>(wfr + ((ct * 4 + s) * 64 + lane) * 16);
}

// ---- input patch.  patch[r][c] = bf16(x[n][row0 + r][c - pl]) (zero outside the image), r < nprow, c < pld.
// Fast form (g.fast: D and pl even, pld <= 128, nprow <= 4*ITER): a thread owns one column pair and ITER rows, all its 8-byte
// loads in flight before the first LDS store.
template <int ITER> struct XStage { float2 v[ITER]; };
template <int ITER>
__device__ __forceinline__ void stem_x_load(XStage<ITER>& xs, const float* __restrict__ x, const StemG& g, int n, int row0, int nprow, int tid) {
    const float* xb = x + (size_t)n * g.T * g.D;
    const int c = 2 * (tid & 63), w0 = c - g.pl;
    const bool cok = c < g.pld && w0 >= 0 && w0 < g.D;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int r = (tid >> 6) + 4 * i, hi = row0 + r;
        const bool ok = cok && r < nprow && hi >= 0 && hi < g.T;
        const float2 v = *reinterpret_cast<const float2*>(xb + (size_t)(ok ? hi : 0) * g.D + (ok ? w0 : 0));
        xs.v[i] = ok ? v : make_float2(0.f, 0.f);
    }
}
// two neighbouring input values into the patch: the 2-byte pair; SAR_X3: + the pair of the lo plane and the fp16 pair of the third plane
__device__ __forceinline__ void stem_put2(u16* dst, int ppe, float a, float b) {
    const unsigned hw = pack2(a, b);
    *reinterpret_cast<unsigned*>(dst) = hw;
    if constexpr (X3) {
        *reinterpret_cast<unsigned*>(dst + ppe) = pack2(a - h_lo(hw), b - h_hi(hw));
        *reinterpret_cast<unsigned*>(dst + 2 * ppe) = wpack2(a, b);
    }
}
template <int ITER>
__device__ __forceinline__ void stem_x_store(const XStage<ITER>& xs, u16* patch, const StemG& g, int nprow, int tid) {
    const int c = 2 * (tid & 63);
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int r = (tid >> 6) + 4 * i;
        if (r < nprow && c < g.pld) stem_put2(patch + r * g.pld + c, g.ppe, xs.v[i].x, xs.v[i].y);
    }
}
// general form: any D / padding, two columns per thread, one row group at a time
__device__ __noinline__ void stem_fill_slow(u16* patch, int ppe, const float* __restrict__ x, int T, int D, int pl, int pld, int n, int row0, int nprow, int tid) {
    const float* xb = x + (size_t)n * T * D;
    const int cp = tid & 63;
    for (int r = tid >> 6; r < nprow; r += 4) {
        const int hi = row0 + r;
        const bool rok = hi >= 0 && hi < T;
        const float* xr = xb + (size_t)(rok ? hi : 0) * D;
        for (int c = 2 * cp; c < pld; c += 128) {
            const int w0 = c - pl, w1 = w0 + 1;
            const bool ok0 = rok && w0 >= 0 && w0 < D, ok1 = rok && w1 >= 0 && w1 < D;
            const float v0 = xr[ok0 ? w0 : 0], v1 = xr[ok1 ? w1 : 0];
            stem_put2(patch + r * pld + c, ppe, ok0 ? v0 : 0.f, ok1 ? v1 : 0.f);
        }
    }
}
// both: kernel fragments and patch of one tile into LDS; `mid` runs between the loads and the stores (the BatchNorm prologue)
template <int NT, int ITER, typename MID>
__device__ __forceinline__ void stem_stage(unsigned char* wfr, u16* patch, const float* __restrict__ x, const float* __restrict__ w, const StemG& g,
                                           int n, int row0, int nprow, int tid, bool with_w, MID mid) {
    WStage<NT> ws; XStage<ITER> xs;
    const bool fast = g.fast && nprow <= 4 * ITER;                                  // block-uniform
    if (with_w) stem_w_load<NT>(ws, w, tid);
    if (fast) stem_x_load<ITER>(xs, x, g, n, row0, nprow, tid);
    mid();
    if (with_w) stem_w_store<NT>(ws, wfr, tid);
    if (fast) stem_x_store<ITER>(xs, patch, g, nprow, tid);
    else stem_fill_slow(patch, g.ppe, x, g.T, g.D, g.pl, g.pld, n, row0, nprow, tid);
}

// One 32-pixel column block: acc[ct] += W(ct) * patches for ct in [CT0, CT0 + NC); boff = (2*ty + h) * pld + 2*tx (bf16 index of
// the lane's first element); lanes of pixels past the tile multiply zeros
template <int NT, int CT0, int NC>
__device__ __forceinline__ void stem_mtile(const h16x8 (&wa)[NT][4], const unsigned char* wfr, int lane, const u16* patch, int ppe, int pld, unsigned boff, bool valid,
                                           f32x16 (&acc)[NC], h16x8 (&bfr)[4]) {
    h16x8 bl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const unsigned* q = reinterpret_cast<const unsigned*>(patch + boff + 2 * s * pld);
        const uint4 v = make_uint4(q[0], q[1], q[2], q[3]);
        bfr[s] = __builtin_bit_cast(h16x8, valid ? v : make_uint4(0u, 0u, 0u, 0u));
        if constexpr (X3) {
            const unsigned* ql = reinterpret_cast<const unsigned*>(patch + ppe + boff + 2 * s * pld);
            const uint4 vl = make_uint4(ql[0], ql[1], ql[2], ql[3]);
            bl[s] = __builtin_bit_cast(h16x8, valid ? vl : make_uint4(0u, 0u, 0u, 0u));
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if constexpr (X3) {                                      // small terms first
                const h16x8 wl = *reinterpret_cast<const h16x8*>(wfr + NT * 4096 + (((CT0 + c) * 4 + s) * 64 + lane) * 16);
                acc[c] = MFMA16(wl, bfr[s], acc[c], 0, 0, 0);
                acc[c] = MFMA16(wa[CT0 + c][s], bl[s], acc[c], 0, 0, 0);
            }
            acc[c] = MFMA16(wa[CT0 + c][s], bfr[s], acc[c], 0, 0, 0);
        }
}

// ---------------------------------------------------------------------------------------------------- forward pass 1: statistics
template <int NT>
__global__ __launch_bounds__(256) void stem16_stats_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ bins, const StemG g, int det) {
    constexpr int C0 = NT * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wfr = smem;                                                       // [NPL][NT*4][64][16 B]
    u16* patch = reinterpret_cast<u16*>(smem + NPL * NT * 4096);
    float* red = reinterpret_cast<float*>(smem + NPL * NT * 4096 + g.patch_bytes);   // [4 waves x 2 lane rows][2][C0]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int n = blockIdx.x / g.tiles, tr = blockIdx.x % g.tiles;
    const int r0 = tr * g.TR, nr = min(g.TR, g.Ho - r0), npix = nr * g.Wo;
    TSTAMP(0); TSTAMP(1);
    stem_stage<NT, 6>(wfr, patch, x, w, g, n, 2 * r0 - g.pt, 2 * (nr - 1) + 8, tid, true, [] {});
    __syncthreads();
    TSTAMP(2);
    h16x8 wa[NT][4];
    stem_w_frags<NT>(wa, wfr, lane);
    float s1[NT][16], s2[NT][16];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1[ct][r] = 0.f; s2[ct][r] = 0.f; }
    for (int mt = wave; mt * 32 < npix; mt += 4) {
        const int p = mt * 32 + (lane & 31);
        const unsigned pc = (unsigned)min(p, npix - 1), ty = udiv(pc, g.mg_wo, (unsigned)g.Wo), tx = pc - ty * g.Wo;
        f32x16 acc[NT]; h16x8 bfr[4];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        stem_mtile<NT, 0, NT>(wa, wfr, lane, patch, g.ppe, g.pld, (2 * ty + h) * g.pld + 2 * tx, p < npix, acc, bfr);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s1[ct][r] += acc[ct][r]; s2[ct][r] = fmaf(acc[ct][r], acc[ct][r], s2[ct][r]); }
    }
    TSTAMP(3);
    // sum over the 32 pixels (lanes) that share h: four DPP adds give every lane the total of its row of 16 lanes (__shfl_xor is
    // ds_bpermute on this target: 640 of them per wave were two thirds of the block's life); the two rows of an h meet in LDS
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = row16_sum(s1[ct][r]), b = row16_sum(s2[ct][r]);
            if ((lane & 15) == 0) {
                const int ch = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, slot = wave * 2 + ((lane >> 4) & 1);
                red[(slot * 2 + 0) * C0 + ch] = a; red[(slot * 2 + 1) * C0 + ch] = b;
            }
        }
    __syncthreads();
    TSTAMP(4);
    if (tid < 2 * C0) {
        const int which = tid / C0, c = tid % C0;
        float S1 = 0.f, S2 = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) { S1 += red[(wv * 2 + 0) * C0 + c]; S2 += red[(wv * 2 + 1) * C0 + c]; }
        // the sums are of the raw products; the BatchNorm input is y = conv + bias
        const float b = bias[c], np = (float)npix;
        const float out = which == 0 ? S1 + np * b : S2 + 2.f * b * S1 + np * b * b;
        // det: a block is one tile of ONE image and sums it in a fixed order, so its partial is a function of the image alone; converted
        // to fixed point, the integer sum over blocks is independent of their order and of the batch they sit in
        if (det) atomicAdd(reinterpret_cast<unsigned long long*>(bins) + ((blockIdx.x % NB) * 2 + which) * C0 + c, (unsigned long long)fix28(out));
        else atomicAdd(bins + ((blockIdx.x % NB) * 2 + which) * C0 + c, out);
    }
    TSTAMP(5); TSTAMP(15);
}

// ---------------------------------------------------------------------------------------------------- forward pass 2: BN + ReLU + max-pool
// A tile = TR pooled rows x the full pooled width; the conv rows its windows touch (2*TR + 1) are recomputed, activated and kept
// as an fp32 LDS tile [pixel][32 channels] (one pass per 32 channels); pooling reads the nine window positions from there.
// Non-negative floats order like unsigned integers, so (bits & ~15 | 15 - slot) under max() yields the maximum AND its first window
// slot in one operation (values closer than 2^-19 relative count as equal).  The comparison is in fp32 on purpose: a bf16 tile made
// ~1 % of the windows pick a neighbour of the true maximum, a 7 % error of the kernel gradient against the float64 oracle.
template <int NT>
__global__ __launch_bounds__(256) void stem16_pool_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          const BnP bn, u16* __restrict__ pooled, u16* __restrict__ pooled_h, uint8_t* __restrict__ idx, const StemG g) {
    // (SAR_X3: `pooled` is a plane pair, pooled_h -- nullable -- the same map as ONE fp16 tensor: the operand of the first block's weight gradients)
    constexpr int C0 = NT * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* cst = reinterpret_cast<float*>(smem);                                     // [4][C0]: scale, shift (+ scale*bias), mean, rstd
    unsigned char* wfr = smem + 4 * C0 * 4;
    u16* patch = reinterpret_cast<u16*>(wfr + NPL * NT * 4096);
    unsigned char* zt = wfr + NPL * NT * 4096 + g.patch_bytes;                       // [pixel][ZP bytes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int n = blockIdx.x / g.tiles, tile = blockIdx.x % g.tiles;
    const int i0 = tile * g.TR, npr = min(g.TR, g.Hp - i0);
    const int r_lo = max(2 * i0 - g.ppt, 0), r_hi = min(2 * (i0 + npr - 1) - g.ppt + 3, g.Ho);
    const int nr = r_hi - r_lo, npix = nr * g.Wo;
    TSTAMP(0); TSTAMP(1);
    stem_stage<NT, 6>(wfr, patch, x, w, g, n, 2 * r_lo - g.pt, 2 * (nr - 1) + 8, tid, true, [&] {
        bn_prologue(bn, 1, C0, 0, C0, cst, C0, blockIdx.x == 0, 256);
        for (int i = tid; i < C0; i += 256) cst[C0 + i] += cst[i] * bias[i];         // same thread wrote both entries
    });
    __syncthreads();
    TSTAMP(2);
    h16x8 wa[NT][4];
    stem_w_frags<NT>(wa, wfr, lane);
    const int nitems = npr * g.Wp * 8;
    auto pass = [&](auto ctc) __attribute__((always_inline)) {
        constexpr int CT = decltype(ctc)::value;
        for (int mt = wave; mt * 32 < npix; mt += 4) {
            const int p = mt * 32 + (lane & 31);
            const bool valid = p < npix;
            const unsigned pc = (unsigned)min(p, npix - 1), ty = udiv(pc, g.mg_wo, (unsigned)g.Wo), tx = pc - ty * g.Wo;
            f32x16 acc[1]; h16x8 bfr[4];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
            stem_mtile<NT, CT, 1>(wa, wfr, lane, patch, g.ppe, g.pld, (2 * ty + h) * g.pld + 2 * tx, valid, acc, bfr);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int cl = 8 * gq + 4 * h, ch0 = CT * 32 + cl;
                const float4 sc = *reinterpret_cast<const float4*>(cst + ch0), sh = *reinterpret_cast<const float4*>(cst + C0 + ch0);
                float4 z;
                z.x = fmaxf(fmaf(acc[0][gq * 4 + 0], sc.x, sh.x), 0.f); z.y = fmaxf(fmaf(acc[0][gq * 4 + 1], sc.y, sh.y), 0.f);
                z.z = fmaxf(fmaf(acc[0][gq * 4 + 2], sc.z, sh.z), 0.f); z.w = fmaxf(fmaf(acc[0][gq * 4 + 3], sc.w, sh.w), 0.f);
                if (valid) *reinterpret_cast<float4*>(zt + (size_t)p * ZP + cl * 4) = z;
            }
        }
        __syncthreads();
        TSTAMP(3 + 2 * CT);
        for (int it = tid; it < nitems; it += 256) {
            const int g4 = it & 7, pp = it >> 3, pj = pp % g.Wp, pi = pp / g.Wp;
            unsigned k0 = 0u, k1 = 0u, k2 = 0u, k3 = 0u;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int r = 2 * (i0 + pi) - g.ppt + kh, c = 2 * pj - g.ppl + kw;
                    const bool ok = r >= r_lo && r < r_hi && c >= 0 && c < g.Wo;
                    const uint4 q = *reinterpret_cast<const uint4*>(zt + (size_t)(ok ? (r - r_lo) * g.Wo + c : 0) * ZP + g4 * 16);
                    const unsigned code = 15u - (unsigned)(kh * 3 + kw), m = ok ? 0xfffffff0u : 0u, cd = ok ? code : 0u;
                    k0 = max(k0, (q.x & m) | cd); k1 = max(k1, (q.y & m) | cd); k2 = max(k2, (q.z & m) | cd); k3 = max(k3, (q.w & m) | cd);
                }
            const size_t o = (((size_t)n * g.Hp + i0 + pi) * g.Wp + pj) * C0 + CT * 32 + g4 * 4;
            const float v0 = __uint_as_float(k0 & ~15u), v1 = __uint_as_float(k1 & ~15u), v2 = __uint_as_float(k2 & ~15u), v3 = __uint_as_float(k3 & ~15u);
            if constexpr (X3) {                                                      // the pooled map is an fp32 tensor (trunk16.h)
                *reinterpret_cast<uint4*>(reinterpret_cast<float*>(pooled) + o) = make_uint4(k0 & ~15u, k1 & ~15u, k2 & ~15u, k3 & ~15u);
                if (pooled_h != nullptr) *reinterpret_cast<uint2*>(pooled_h + o) = make_uint2(wpack2(v0, v1), wpack2(v2, v3));
            } else {
                *reinterpret_cast<uint2*>(pooled + o) = make_uint2(pack2(v0, v1), pack2(v2, v3));
            }
            *reinterpret_cast<unsigned*>(idx + o) = (15u - (k0 & 15u)) | ((15u - (k1 & 15u)) << 8) | ((15u - (k2 & 15u)) << 16) | ((15u - (k3 & 15u)) << 24);
        }
    };
    pass(std::integral_constant<int, 0>{});
    TSTAMP(4);
    if constexpr (NT == 2) {
        __syncthreads();                                                             // the pooling reads of pass 0 are done
        pass(std::integral_constant<int, 1>{});
    }
    TSTAMP(6); TSTAMP(15);
}

// ---------------------------------------------------------------------------------------------------- backward pass 1: the two sums
// conv output at one position, for the degenerate-gamma fallback below (never taken by a trained or freshly initialised model)
__device__ __noinline__ float stem_point(const float* __restrict__ x, const float* __restrict__ w, float b, int T, int D, int pt, int pl, int C0, int n, int r, int c, int ch) {
    float y = 0.f;
    for (int ky = 0; ky < 7; ++ky)
        for (int kx = 0; kx < 7; ++kx) {
            const int hi = 2 * r - pt + ky, wi = 2 * c - pl + kx;
            if (hi < 0 || hi >= T || wi < 0 || wi >= D) continue;
            const float xv = h_lo(pack2(x[((size_t)n * T + hi) * D + wi], 0.f)), wv = h_lo(pack2(w[(ky * 7 + kx) * C0 + ch], 0.f));
            y = fmaf(xv, wv, y);
        }
    return y + b;
}

// bbins[bin][0][c] += sum dz, bbins[bin][1][c] += sum dz * xhat over the conv output, where dz is the pooled gradient routed to
// the argmax position of every window whose maximum is positive (ReLU active).  There z = gamma*xhat + beta equals the pooled
// value, so xhat = (pooled - beta) / gamma: one pass over the pooled tensors, no full-resolution operand.
// PH (SAR_X3, round 5): `pooled` is the fp16 copy of the pooled map the forward pass also writes (pooled_h: the first residual block's weight-gradient operand)
// -- these are BatchNorm-backward SUMS, gradient-class like everything else the mixed backward makes from fp16 copies, and the fp32 map was 33 of
// this pass's 47 MB.
template <bool PH = false>
__global__ __launch_bounds__(256) void stem16_bwd_reduce_kernel(const u16* __restrict__ gp, const u16* __restrict__ pooled, const uint8_t* __restrict__ idx,
                                                                const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ saved,
                                                                float* __restrict__ bbins, int* __restrict__ counter, const StemG g, unsigned npix) {
    __shared__ float r1[256 * 8], r2[256 * 8];
    if (blockIdx.x == 0 && threadIdx.x == 0) *counter = 0;                           // the tile counter of stem16_wgrad_kernel
    const int C0 = g.C0, tpr = C0 / 8;
    const int tid = threadIdx.x, cv = tid % tpr, pg = tid / tpr, npg = 256 / tpr;
    float be[8], ig[8]; bool okc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float gm = gamma[cv * 8 + k];
        be[k] = beta[cv * 8 + k]; okc[k] = fabsf(gm) > 1e-6f; ig[k] = okc[k] ? 1.f / gm : 0.f;
    }
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
    for (unsigned pix = blockIdx.x * npg + pg; pix < npix; pix += gridDim.x * npg) {
        const size_t o = (size_t)pix * C0 + cv * 8;
        float a[8], gg[8];
        if constexpr (X3) {                                                          // pooled: an fp32 tensor; gp: fp16, scaled (round 5: the mixed backward)
            if constexpr (PH) {
                const uint4 pq = *reinterpret_cast<const uint4*>(pooled + o);
                a[0] = w_lo(pq.x); a[1] = w_hi(pq.x); a[2] = w_lo(pq.y); a[3] = w_hi(pq.y); a[4] = w_lo(pq.z); a[5] = w_hi(pq.z); a[6] = w_lo(pq.w); a[7] = w_hi(pq.w);
            } else f32x8_of(ld16f(pooled, o), ld16f(pooled, o + 4), a);
            const uint4 gq = *reinterpret_cast<const uint4*>(gp + o);
            gg[0] = w_lo(gq.x); gg[1] = w_hi(gq.x); gg[2] = w_lo(gq.y); gg[3] = w_hi(gq.y); gg[4] = w_lo(gq.z); gg[5] = w_hi(gq.z); gg[6] = w_lo(gq.w); gg[7] = w_hi(gq.w);
        } else {
            unpack8(*reinterpret_cast<const uint4*>(pooled + o), a); unpack8(*reinterpret_cast<const uint4*>(gp + o), gg);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (a[k] > 0.f) {
                float xh;
                if (okc[k]) xh = (a[k] - be[k]) * ig[k];
                else {
                    const int pj = pix % g.Wp, pi = (pix / g.Wp) % g.Hp, n = pix / (g.Wp * g.Hp), slot = idx[o + k], ch = cv * 8 + k;
                    const float y = stem_point(x, w, bias[ch], g.T, g.D, g.pt, g.pl, C0, n, 2 * pi - g.ppt + slot / 3, 2 * pj - g.ppl + slot % 3, ch);
                    xh = (y - saved[ch]) * saved[C0 + ch];
                }
                s1[k] += gg[k]; s2[k] = fmaf(gg[k], xh, s2[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { r1[tid * 8 + k] = s1[k]; r2[tid * 8 + k] = s2[k]; }
    __syncthreads();
    if (tid < 2 * C0) {
        const int which = tid / C0, c = tid % C0;
        const float* rr = which ? r2 : r1;
        float s = 0.f;
        for (int q = 0; q < npg; ++q) s += rr[(q * tpr + c / 8) * 8 + (c & 7)];
        atomicAdd(bbins + ((blockIdx.x % NB) * 2 + which) * C0 + c, s);
    }
}

// ---------------------------------------------------------------------------------------------------- backward pass 2: kernel gradient
// part[block][ky*8 + kx][c] = this block's share of dW.  A tile = TR conv rows x the full width, 32 channels per pass:
//  scatter : every pooled window whose maximum is positive adds its gradient to the conv pixel its argmax slot names -- one
//            packed-bf16 LDS atomic (ds_pk_add_bf16) into the tile's dz image [pixel][32 ch].  (The first form gathered instead: every
//            pixel tested its <= 4 candidate windows per channel -- 5 VALU instructions per candidate and channel, ~900 per
//            32-pixel block, 4 000 cycles of a 5 000-cycle block with one wave per SIMD; tools/stem_timing.py.)
//  m-tiles : per 32-pixel column block a wave recomputes the conv, turns it into dy = A*dz + E*acc + F per channel (A = scale,
//            E = -scale*c3*rstd, F = -scale*(c2 + c3*(bias - mean)*rstd): BatchNorm backward with xhat = (acc + bias - mean)*rstd
//            folded in), clears the dz entries it read, writes dy and its own B fragments (= the im2col rows of its pixels) as
//            pixel-major bf16 images into wave-private LDS and multiplies them through transpose-reads: no barrier inside.
// Tiles are handed out through an atomic counter (work[0], zeroed by the reduce pass in front of this kernel): blocks stay
// resident, keep their accumulators, and the chip stays balanced whatever the tile count.
constexpr int DZP = 64 + 8;     // bytes per pixel row of the dz image (32 bf16 + 8: the 8-byte reads of 16 pixels hit 32 different banks)

template <int NT>
__global__ __launch_bounds__(256, 2) void stem16_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                              const u16* __restrict__ gp, const u16* __restrict__ pooled, const uint8_t* __restrict__ idx,
                                                              const BnP bn, int* __restrict__ counter, float* __restrict__ part, const StemG g, int ntiles,
                                                              int dz_bytes, float gs) {
    // SAR_X3: pooled is an fp32 tensor, gp the SCALED fp16 gradient of the mixed backward (round 5; until then fp32 in true units, scaled by gs
    // here); the conv is recomputed in split precision; the scattered gradient image, dy and the im2col image are fp16 (w16)
    constexpr int C0 = NT * 32, SCR = 3 * PLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_tile;
    float* cst = reinterpret_cast<float*>(smem);                                     // [6][C0] from the prologue, then [3][C0]: A, E, F
    float* aef = cst + 6 * C0;
    unsigned char* wfr = smem + 9 * C0 * 4;
    u16* patch = reinterpret_cast<u16*>(wfr + NPL * NT * 4096);
    unsigned char* dz = wfr + NPL * NT * 4096 + g.patch_bytes;                       // [pixel][DZP]
    unsigned char* scr0 = dz + dz_bytes + 256;                                       // [4 waves][SCR]: P planes 0, 1, dy plane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, q = lane & 31;
    unsigned char* scr = scr0 + wave * SCR;
    h16x8 wa[NT][4];
    f32x16 accw[2][NT];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) accw[tp][ct][r] = 0.f;
    const int gi = lane & 15, gg = lane >> 4, cb = gg & 1, kh2 = gg >> 1;
    const unsigned colb = (16 * cb + 4 * (gi & 3)) * 2, prb = 8 * kh2 + (gi >> 2);
    TSTAMP(0); TSTAMP(1);

    // once per block: kernel fragments, BatchNorm constants, a clean dz image
    {
        WStage<NT> ws;
        stem_w_load<NT>(ws, w, tid);
        bn_prologue(bn, 2, C0, 0, C0, cst, C0, blockIdx.x == 0, 256);
        for (int i = tid; i < C0; i += 256) {
            const float scale = cst[i], mean = cst[2 * C0 + i], rstd = cst[3 * C0 + i], c2 = cst[4 * C0 + i], c3 = cst[5 * C0 + i];
            const float xb = (bias[i] - mean) * rstd, Cc = -scale * c3;
            aef[i] = scale; aef[C0 + i] = Cc * rstd * gs; aef[2 * C0 + i] = (-scale * c2 + Cc * xb) * gs;
        }
        stem_w_store<NT>(ws, wfr, tid);
        for (int i = tid * 16; i < dz_bytes; i += 256 * 16) *reinterpret_cast<uint4*>(dz + i) = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        stem_w_frags<NT>(wa, wfr, lane);
    }
    TSTAMP(2);
#ifdef SAR_TIMING
    unsigned long long t_stage = 0, t_mt = 0, t_n = 0, t_a, t_b;
#endif
    // (Drawing the ticket of the NEXT tile while the current one is worked on -- to take the returning atomic's round trip off the
    // front of every tile -- was measured: 84 -> 93 us.  A block then holds a tile it will only start a tile-time later, and with ~4
    // tiles per block that costs more balance at the end than the round trips cost in front.)
    for (;;) {
        if (tid == 0) s_tile = atomicAdd(counter, 1);
        __syncthreads();                                                             // also: the previous tile's readers are done, its dz entries cleared
        const int tile = s_tile;
        if (tile >= ntiles) break;
        const int n = tile / g.tiles, tr = tile % g.tiles;
        const int r0 = tr * g.TR, nr = min(g.TR, g.Ho - r0), npix = nr * g.Wo;
        const int i_lo = max((r0 + g.ppt - 1) >> 1, 0), i_hi = min((r0 + nr - 1 + g.ppt) >> 1, g.Hp - 1), nwr = i_hi - i_lo + 1;
        const int nitems = nwr * g.Wp * 4;                                           // (window, 8 of the pass's 32 channels)
        auto pass = [&](auto ctc) __attribute__((always_inline)) {
            constexpr int CT = decltype(ctc)::value;
#ifdef SAR_TIMING
            t_a = __builtin_amdgcn_s_memtime();
#endif
            auto scatter = [&]() __attribute__((always_inline)) {
                uint4 pv[2], gv[2], gl[2]; uint2 iv[2];
                // the gradient pair of channel k as the dz image takes it: the raw 16 bits; SAR_X3: fp16((hi + lo) * gs)
                // (SAR_X3: gp and pooled are fp32 tensors: hq / lq = elements 0-3 / 4-7 of the eight)
                auto gbits = [&](const uint4 hq, const uint4 lq, int k) __attribute__((always_inline)) -> unsigned {
                    const unsigned hs[4] = {hq.x, hq.y, hq.z, hq.w}, sh = 16 * (k & 1);
                    return (hs[k >> 1] >> sh) & 0xffffu;                              // (SAR_X3, round 5: gp is the scaled fp16 tensor the image takes as it is)
                };
                // which of the eight pooled values are positive (>= 0 always: non-zero magnitude bits), one bit each
                auto pmask = [&](const uint4 a, const uint4 b) __attribute__((always_inline)) -> unsigned {
                    if constexpr (X3) {
                        const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                        unsigned m = 0u;
#pragma unroll
                        for (int k = 0; k < 8; ++k) m |= ((w[k] & 0x7fffffffu) != 0u ? 1u : 0u) << k;
                        return m;
                    } else {
                        const unsigned w[4] = {a.x, a.y, a.z, a.w};
                        unsigned m = 0u;
#pragma unroll
                        for (int k = 0; k < 8; ++k) m |= ((((w[k >> 1] >> (16 * (k & 1))) & 0x7fffu) != 0u) ? 1u : 0u) << k;
                        return m;
                    }
                };
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int it = min(tid + 256 * u, nitems - 1), cg = it & 3, pp = it >> 2;
                    const size_t o = (((size_t)n * g.Hp + i_lo) * g.Wp + pp) * C0 + CT * 32 + cg * 8;
                    if constexpr (X3) {
                        const uint4 pa = ld16f(pooled, o), pb = ld16f(pooled, o + 4);
                        gv[u] = *reinterpret_cast<const uint4*>(gp + o); gl[u] = gv[u];
                        pv[u] = make_uint4(pmask(pa, pb), 0u, 0u, 0u);
                    } else {
                        pv[u] = *reinterpret_cast<const uint4*>(pooled + o); gv[u] = *reinterpret_cast<const uint4*>(gp + o);
                        gl[u] = gv[u];
                        pv[u] = make_uint4(pmask(pv[u], pv[u]), 0u, 0u, 0u);
                    }
                    iv[u] = *reinterpret_cast<const uint2*>(idx + o);
                }
                for (int it0 = tid + 512; it0 < nitems; it0 += 256) {               // tiles with more than 512 items (wide inputs): one at a time
                    const int cg = it0 & 3, pp = it0 >> 2;
                    const size_t o = (((size_t)n * g.Hp + i_lo) * g.Wp + pp) * C0 + CT * 32 + cg * 8;
                    uint4 g1, g1l; unsigned pm;
                    if constexpr (X3) { g1 = *reinterpret_cast<const uint4*>(gp + o); g1l = g1; pm = pmask(ld16f(pooled, o), ld16f(pooled, o + 4)); }
                    else { const uint4 p1 = *reinterpret_cast<const uint4*>(pooled + o); g1 = *reinterpret_cast<const uint4*>(gp + o); g1l = g1; pm = pmask(p1, p1); }
                    const uint2 i1 = *reinterpret_cast<const uint2*>(idx + o);
                    const int wi = pp / g.Wp, wj = pp - wi * g.Wp;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const unsigned sh = 16 * (k & 1), pz = (pm >> k) & 1u, gz = gbits(g1, g1l, k);
                        const int slot = ((k < 4 ? i1.x : i1.y) >> (8 * (k & 3))) & 255, kh = (slot * 11) >> 5, kw = slot - 3 * kh;
                        const int r = 2 * (i_lo + wi) - g.ppt + kh - r0, c = 2 * wj - g.ppl + kw;
                        if (pz != 0u && r >= 0 && r < nr) {
                            lds_pk_add_w(dz + (size_t)(r * g.Wo + c) * DZP + (cg * 8 + (k & ~1)) * 2, gz << sh);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int it = tid + 256 * u, cg = it & 3, pp = it >> 2;
                    const int wi = (int)udiv((unsigned)pp, g.mg_wp, (unsigned)g.Wp), wj = pp - wi * g.Wp;
                    const unsigned pm = pv[u].x;                                    // (the positivity bits of the eight pooled values)
                    const int rb = 2 * (i_lo + wi) - g.ppt - r0, cbase = 2 * wj - g.ppl;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        // pooled >= 0 always (ReLU): positive <=> non-zero magnitude bits.  (A branch-free form -- misses add 0 to a dummy
                        // word -- doubled the phase: it is the LDS atomics that cost, not the branches.)
                        const unsigned sh = 16 * (k & 1), pz = (pm >> k) & 1u, gz = gbits(gv[u], gl[u], k);
                        const int slot = ((k < 4 ? iv[u].x : iv[u].y) >> (8 * (k & 3))) & 255, kh = (slot * 11) >> 5, kw = slot - 3 * kh;
                        const int r = rb + kh, c = cbase + kw;
                        if (it < nitems && pz != 0u && r >= 0 && r < nr) {
                            lds_pk_add_w(dz + (size_t)(r * g.Wo + c) * DZP + (cg * 8 + (k & ~1)) * 2, gz << sh);
                        }
                    }
                }
            };
            if (CT == 0) stem_stage<NT, 6>(wfr, patch, x, w, g, n, 2 * r0 - g.pt, 2 * (nr - 1) + 8, tid, false, scatter);
            else scatter();
            __syncthreads();
#ifdef SAR_TIMING
            t_b = __builtin_amdgcn_s_memtime(); t_stage += t_b - t_a;
#endif
            // 10 column blocks over 4 waves: the waves that take the odd ones out rotate with the pass
            for (int mt = (wave + 2 * CT) & 3; mt * 32 < npix; mt += 4) {
                const int p = mt * 32 + q;
                const bool valid = p < npix;
                const unsigned pc = (unsigned)min(p, npix - 1), ty = udiv(pc, g.mg_wo, (unsigned)g.Wo), tx = pc - ty * g.Wo;
                f32x16 acc[1]; h16x8 bfr[4];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
                stem_mtile<NT, CT, 1>(wa, wfr, lane, patch, g.ppe, g.pld, (2 * ty + h) * g.pld + 2 * tx, valid, acc, bfr);
                // the lane's B fragments ARE its pixel's im2col row (taps (2s+h)*8 .. +8): plane (2s+h)/4, 16 bytes at ((2s+h)%4)*16
                // (SAR_X3: the same four dwords of the patch's fp16 plane)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    if constexpr (X3) {
                        const unsigned* qw = reinterpret_cast<const unsigned*>(patch + 2 * g.ppe + (2 * ty + h) * g.pld + 2 * tx + 2 * s2 * g.pld);
                        const uint4 vw = make_uint4(qw[0], qw[1], qw[2], qw[3]);
                        *reinterpret_cast<uint4*>(scr + ((2 * s2 + h) >> 2) * PLANE + q * IP + ((2 * s2 + h) & 3) * 16) = valid ? vw : make_uint4(0u, 0u, 0u, 0u);
                    } else *reinterpret_cast<h16x8*>(scr + ((2 * s2 + h) >> 2) * PLANE + q * IP + ((2 * s2 + h) & 3) * 16) = bfr[s2];
                }
                unsigned char* dzp = dz + (size_t)pc * DZP + 8 * h;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int cl = 8 * gq + 4 * h, ch0 = CT * 32 + cl;
                    uint2 dv = *reinterpret_cast<const uint2*>(dzp + 16 * gq);
                    if (valid) *reinterpret_cast<uint2*>(dzp + 16 * gq) = make_uint2(0u, 0u);       // leave the image clean for the next scatter
                    else dv = make_uint2(0u, 0u);
                    const float4 A4 = *reinterpret_cast<const float4*>(aef + ch0), E4 = *reinterpret_cast<const float4*>(aef + C0 + ch0),
                                 F4 = *reinterpret_cast<const float4*>(aef + 2 * C0 + ch0);
                    // lanes past the tile: acc = 0 and dz = 0, so F alone has to be masked
                    const float d0 = fmaf(A4.x, w_lo(dv.x), fmaf(E4.x, acc[0][gq * 4 + 0], valid ? F4.x : 0.f));
                    const float d1 = fmaf(A4.y, w_hi(dv.x), fmaf(E4.y, acc[0][gq * 4 + 1], valid ? F4.y : 0.f));
                    const float d2 = fmaf(A4.z, w_lo(dv.y), fmaf(E4.z, acc[0][gq * 4 + 2], valid ? F4.z : 0.f));
                    const float d3 = fmaf(A4.w, w_hi(dv.y), fmaf(E4.w, acc[0][gq * 4 + 3], valid ? F4.w : 0.f));
                    *reinterpret_cast<uint2*>(scr + 2 * PLANE + q * IP + cl * 2) = make_uint2(wpack2(d0, d1), wpack2(d2, d3));
                }
                // the wave's own LDS stores are ordered in front of its own reads (same queue); keep the compiler from reordering them
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const unsigned o0 = (16 * ks + prb) * IP + colb, o1 = o0 + 4 * IP;
                    w16x8 Af[2], Bf;
#pragma unroll
                    for (int tp = 0; tp < 2; ++tp) {
                        const s16x4 lo = lds_tr(scr + tp * PLANE, o0), hi = lds_tr(scr + tp * PLANE, o1);
                        Af[tp] = __builtin_bit_cast(w16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                    {
                        const s16x4 lo = lds_tr(scr + 2 * PLANE, o0), hi = lds_tr(scr + 2 * PLANE, o1);
                        Bf = __builtin_bit_cast(w16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
#pragma unroll
                    for (int tp = 0; tp < 2; ++tp) accw[tp][CT] = MFMA16W(Af[tp], Bf, accw[tp][CT], 0, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // the transpose-reads are done before the images are rewritten
            }
#ifdef SAR_TIMING
            t_mt += __builtin_amdgcn_s_memtime() - t_b;
#endif
        };
        pass(std::integral_constant<int, 0>{});
        if constexpr (NT == 2) {
            __syncthreads();                                                         // pass 0's readers are done: the dz image is clean again
            pass(std::integral_constant<int, 1>{});
        }
#ifdef SAR_TIMING
        ++t_n;
#endif
    }
    TSTAMP(3);
#ifdef SAR_TIMING
    if (g_tbuf_stem && threadIdx.x == 0) { g_tbuf_stem[(long)blockIdx.x * 16 + 8] = t_stage; g_tbuf_stem[(long)blockIdx.x * 16 + 9] = t_mt; g_tbuf_stem[(long)blockIdx.x * 16 + 10] = t_n; }
#endif
    // ---- merge the four waves' tiles in LDS (over patch / dz / scratch: everything is dead), then one coalesced store
    float* ot = reinterpret_cast<float*>(wfr);                                       // [64][C0]
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = tp * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = ct * 32 + q;
                        if (wv == 0) ot[row * C0 + col] = accw[tp][ct][r]; else ot[row * C0 + col] += accw[tp][ct][r];
                    }
        }
        __syncthreads();
    }
    float4* dst = reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 64 * C0);
    for (int i = tid; i < 64 * C0 / 4; i += 256) dst[i] = reinterpret_cast<const float4*>(ot)[i];
    TSTAMP(4); TSTAMP(15);
}

constexpr int WS_SLICES = 8;    // slices of the blocks' parts in the kernel gradient's final sum (stem16_wsum_kernel / stem16_wfin_kernel)

// ---------------------------------------------------------------------------------------------------- backward pass 2, second form (round 6)
// The kernel gradient WITHOUT recomputing the conv.  With P = the im2col rows of the conv pixels (64 taps, ky*8 + kx), dz = the scattered pooled
// gradient and dy = A*dz + E*acc + F (above), acc = P w:
//     dW = P^T dy = A (.) (P^T dz) + E (.) ((P^T P) w) + (P^T 1) (x) F
// -- the dense part of dy is a fixed linear map of P, so its product with P^T is a product with the 64 x 64 GRAM matrix G = P^T P of the input's
// patches, which no channel, no gradient and no BatchNorm constant enters.  stem16_wgrad_kernel spent its m-tiles on exactly that dense part: per
// 32-pixel block and 32-channel pass twelve MFMAs to recompute acc (split precision), sixteen FMAs per lane to form dy, an image write -- to multiply
// a term that is known in closed form.  Here a block accumulates T = P^T dz (the dz image feeds the MFMA as it lies: its transposed reads ARE the B
// fragments) and G (its A fragments are its B fragments: three MFMAs per 16 pixels, no further LDS read), with P^T 1 riding along as column 7 of G (tap
// ky = 0, kx = 7 is a zero row of the kernel: the image carries 1.0 there); stem16_wfin_kernel then forms A (.) T + E (.) (G w) + G[:, 7] (x) F per slice of
// the blocks -- the map is linear, so slices add.  No kernel fragments, no BatchNorm constants, one patch plane, ONE scatter for all channels per tile.
// The products are single-pass on the same 2-byte operands as before (P and dz), the rest fp32: dy is no longer rounded to 2 bytes, G sums 5e5 products.
template <int NT>
__global__ __launch_bounds__(256, 2) void stem16_wgrad2_kernel(const float* __restrict__ x, const u16* __restrict__ gp, const u16* __restrict__ pooled,
                                                               const uint8_t* __restrict__ idx, int* __restrict__ counter, float* __restrict__ part,
                                                               const StemG g, int ntiles, int dz_bytes) {
    constexpr int C0 = NT * 32, DZ2 = C0 * 2 + 8, CG = C0 / 8, LCG = NT == 2 ? 3 : 2;   // dz image: bytes per pixel row, channel groups of eight
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u16* patch = reinterpret_cast<u16*>(smem);                                       // ONE plane of the weight gradient's 2-byte type, [row][pld]
    unsigned char* dz = smem + g.patch_bytes;                                        // [pixel, rounded up to 32][DZ2]
    unsigned char* scr0 = dz + dz_bytes + 256;                                       // [4 waves][2 planes]: the im2col image of a 32-pixel block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, q = lane & 31;
    unsigned char* scr = scr0 + wave * 2 * PLANE;
    f32x16 accw[2][NT], accg[3];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) accw[tp][ct][r] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) accg[k][r] = 0.f;
    }
    const int gi = lane & 15, gg = lane >> 4, cb = gg & 1, kh2 = gg >> 1;
    const unsigned colb = (16 * cb + 4 * (gi & 3)) * 2, prb = 8 * kh2 + (gi >> 2);
    const unsigned one16 = (unsigned)__builtin_bit_cast(unsigned short, (w16)1.f) << 16;   // 1.0 in the high half of a pair
    for (int i = tid * 16; i < dz_bytes; i += 256 * 16) *reinterpret_cast<uint4*>(dz + i) = make_uint4(0u, 0u, 0u, 0u);
    // Tiles are dealt statically, tile = block + k * blocks: with ~3 tiles per block a ticket balances nothing a static deal does not (round 3: drawing
    // the next ticket early cost balance), and a block that knows its next tile requests that tile's operands -- the input rows, the pooled windows'
    // gradient, positivity and slots -- BEFORE it multiplies the current one: the round trip that stood in front of every tile runs under the m-tiles.
    struct TG { int n, r0, nr, npix, i_lo, nitems; };
    auto geo = [&](int tile) __attribute__((always_inline)) {
        TG t;
        t.n = tile / g.tiles; const int tr = tile % g.tiles;
        t.r0 = tr * g.TR; t.nr = min(g.TR, g.Ho - t.r0); t.npix = t.nr * g.Wo;
        t.i_lo = max((t.r0 + g.ppt - 1) >> 1, 0);
        const int i_hi = min((t.r0 + t.nr - 1 + g.ppt) >> 1, g.Hp - 1);
        t.nitems = (i_hi - t.i_lo + 1) * g.Wp * CG;                                  // (window, eight channels)
        return t;
    };
    // which of eight pooled values are positive (>= 0 always: non-zero magnitude bits), one bit each; SAR_X3: `pooled` is the fp32 map
    auto pmask_at = [&](size_t o) __attribute__((always_inline)) -> unsigned {
        unsigned m = 0u;
        if constexpr (X3) {
            const uint4 a = ld16f(pooled, o), b = ld16f(pooled, o + 4);
            const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) m |= ((w[k] & 0x7fffffffu) != 0u ? 1u : 0u) << k;
        } else {
            const uint4 a = *reinterpret_cast<const uint4*>(pooled + o);
            const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) m |= ((((w[k >> 1] >> (16 * (k & 1))) & 0x7fffu) != 0u) ? 1u : 0u) << k;
        }
        return m;
    };
    auto item_off = [&](const TG& t, int it) __attribute__((always_inline)) -> size_t {
        return (((size_t)t.n * g.Hp + t.i_lo) * g.Wp + (it >> LCG)) * C0 + (it & (CG - 1)) * 8;
    };
    constexpr int SU = NT == 2 ? 3 : 2;                                              // items per thread held in registers (a tile of eight rows: ~800 / ~400 items)
    XStage<6> xs; unsigned pm[SU]; uint4 gv[SU]; uint2 iv[SU];
    auto request = [&](const TG& t) __attribute__((always_inline)) {
        stem_x_load<6>(xs, x, g, t.n, 2 * t.r0 - g.pt, 2 * (t.nr - 1) + 8, tid);
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const size_t o = item_off(t, min(tid + 256 * u, t.nitems - 1));
            pm[u] = pmask_at(o); gv[u] = *reinterpret_cast<const uint4*>(gp + o); iv[u] = *reinterpret_cast<const uint2*>(idx + o);
        }
    };
    int tile = blockIdx.x;
    TG t = geo(min(tile, ntiles - 1));
    if (tile < ntiles) request(t);
    while (tile < ntiles) {
        const int n = t.n, r0 = t.r0, nr = t.nr, npix = t.npix, i_lo = t.i_lo, nitems = t.nitems;
        (void)n;
        __syncthreads();                                                             // the previous tile's readers are done, its dz entries cleared
        // ---- stage: the scatter's LDS adds, the patch's stores (their operands were requested a tile ago)
        {
            auto add_item = [&](int it, unsigned pmk, const uint4 gvv, const uint2 ivv) __attribute__((always_inline)) {
                const int cg = it & (CG - 1), pp = it >> LCG;
                const int wi = (int)udiv((unsigned)pp, g.mg_wp, (unsigned)g.Wp), wj = pp - wi * g.Wp;
                const int rb = 2 * (i_lo + wi) - g.ppt - r0, cbase = 2 * wj - g.ppl;
                const unsigned gw[4] = {gvv.x, gvv.y, gvv.z, gvv.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned sh = 16 * (k & 1), gz = (gw[k >> 1] >> sh) & 0xffffu;
                    const int slot = ((k < 4 ? ivv.x : ivv.y) >> (8 * (k & 3))) & 255, kh = (slot * 11) >> 5, kw = slot - 3 * kh;
                    const int r = rb + kh, c = cbase + kw;
                    if (((pmk >> k) & 1u) != 0u && r >= 0 && r < nr)
                        lds_pk_add_w(dz + (size_t)(r * g.Wo + c) * DZ2 + (cg * 8 + (k & ~1)) * 2, gz << sh);
                }
            };
#pragma unroll
            for (int u = 0; u < SU; ++u)
                if (tid + 256 * u < nitems) add_item(tid + 256 * u, pm[u], gv[u], iv[u]);
            for (int it = tid + 256 * SU; it < nitems; it += 256) {                  // wider inputs: one at a time
                const size_t o = item_off(t, it);
                add_item(it, pmask_at(o), *reinterpret_cast<const uint4*>(gp + o), *reinterpret_cast<const uint2*>(idx + o));
            }
            const int c = 2 * (tid & 63), nprow = 2 * (nr - 1) + 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int r = (tid >> 6) + 4 * i;
                if (r < nprow && c < g.pld) *reinterpret_cast<unsigned*>(patch + r * g.pld + c) = wpack2(xs.v[i].x, xs.v[i].y);
            }
        }
        __syncthreads();
        tile += gridDim.x;
        if (tile < ntiles) { t = geo(tile); request(t); }                            // the NEXT tile's operands, under this tile's m-tiles
        // ---- 32-pixel blocks over the four waves
        for (int mt = wave; mt * 32 < npix; mt += 4) {
            const int p = mt * 32 + q;
            const bool valid = p < npix;
            const unsigned pc = (unsigned)min(p, npix - 1), ty = udiv(pc, g.mg_wo, (unsigned)g.Wo), tx = pc - ty * g.Wo;
            // the lane's pixel's im2col row, taps (2s + h) * 8 .. + 8: plane (2s + h) / 4, 16 bytes at ((2s + h) % 4) * 16; tap 7 (ky = 0, kx = 7) carries 1.0
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const unsigned* qw = reinterpret_cast<const unsigned*>(patch + (2 * ty + h) * g.pld + 2 * tx + 2 * s2 * g.pld);
                uint4 vw = make_uint4(qw[0], qw[1], qw[2], qw[3]);
                if (s2 == 0 && h == 0) vw.w = (vw.w & 0xffffu) | one16;
                *reinterpret_cast<uint4*>(scr + ((2 * s2 + h) >> 2) * PLANE + q * IP + ((2 * s2 + h) & 3) * 16) = valid ? vw : make_uint4(0u, 0u, 0u, 0u);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // the wave's own stores are in front of its own reads
            unsigned char* dzb = dz + (size_t)(mt * 32) * DZ2;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned o0 = (16 * ks + prb) * IP + colb, o1 = o0 + 4 * IP;
                const unsigned d0 = (16 * ks + prb) * DZ2 + colb, d1 = d0 + 4 * DZ2;
                w16x8 Af[2], Bd[NT];
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    const s16x4 lo = lds_tr(scr + tp * PLANE, o0), hi = lds_tr(scr + tp * PLANE, o1);
                    Af[tp] = __builtin_bit_cast(w16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) {
                    const s16x4 lo = lds_tr(dzb + ct * 64, d0), hi = lds_tr(dzb + ct * 64, d1);
                    Bd[ct] = __builtin_bit_cast(w16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // leave the image clean for the next tile's scatter: every 8-byte piece of the block's rows was read by exactly one lane
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) {
                    *reinterpret_cast<uint2*>(dzb + ct * 64 + d0) = make_uint2(0u, 0u);
                    *reinterpret_cast<uint2*>(dzb + ct * 64 + d1) = make_uint2(0u, 0u);
                }
#pragma unroll
                for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) accw[tp][ct] = MFMA16W(Af[tp], Bd[ct], accw[tp][ct], 0, 0, 0);
                accg[0] = MFMA16W(Af[0], Af[0], accg[0], 0, 0, 0);
                accg[1] = MFMA16W(Af[0], Af[1], accg[1], 0, 0, 0);
                accg[2] = MFMA16W(Af[1], Af[1], accg[2], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // reads and clears done before the images are rewritten
        }
    }
    // ---- merge the four waves' tiles in LDS (everything is dead), then coalesced stores: part[block] = {T [64][C0], G [64][64] (rows 32-63 x columns 0-31 unset)}
    __syncthreads();
    float* ot = reinterpret_cast<float*>(smem);                                      // [64][C0] | [64][64]
    float* og = ot + 64 * C0;
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
                for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) {
                        const int row = tp * 32 + rr, col = ct * 32 + q;
                        if (wv == 0) ot[row * C0 + col] = accw[tp][ct][r]; else ot[row * C0 + col] += accw[tp][ct][r];
                    }
                const int gr[3] = {rr, rr, 32 + rr}, gc[3] = {q, 32 + q, 32 + q};
#pragma unroll
                for (int k = 0; k < 3; ++k) { if (wv == 0) og[gr[k] * 64 + gc[k]] = accg[k][r]; else og[gr[k] * 64 + gc[k]] += accg[k][r]; }
            }
        }
        __syncthreads();
    }
    float4* dst = reinterpret_cast<float4*>(part + (size_t)blockIdx.x * (64 * C0 + 64 * 64));
    for (int i = tid; i < (64 * C0 + 64 * 64) / 4; i += 256) dst[i] = reinterpret_cast<const float4*>(ot)[i];
}

// dw[(ky*7 + kx)][c] += (A_c T[k][c] + E_c (G w)[k][c] + G[k][7] F_c) * ginv over one slice of the blocks' parts (the map is linear: slices add);
// grid (49 taps, WS_SLICES).  A = scale, E = -scale c3 rstd, F = -scale (c2 + c3 (bias - mean) rstd) from the BatchNorm's backward sums (bn: saved +
// bbins); block (0, 0) owns the BatchNorm's parameter gradients when bn.write, as stem16_wgrad_kernel's block 0 did.
__global__ __launch_bounds__(256) void stem16_wfin_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ w, const float* __restrict__ bias,
                                                          const BnP bn, float* __restrict__ dw, int C0, int* __restrict__ counter) {
    __shared__ float cst[6 * 64], red[256], grow[64], gred[256];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *counter = 0;
    const int t = blockIdx.x, row = (t / 7) * 8 + (t % 7), tid = threadIdx.x;
    const int c = tid % C0, sl = tid / C0, nsl = 256 / C0;
    const size_t ps = (size_t)64 * C0 + 64 * 64;
    const int per = (nblk + WS_SLICES - 1) / WS_SLICES, b0 = blockIdx.y * per, b1 = min(b0 + per, nblk);
    // every request of this block first -- the slice's parts of T[row][c] and G[row][k'], the kernel column of this thread's channel, then (inside
    // bn_prologue) the BatchNorm's sums -- and only then the first use: the block is three round trips if they are taken one after the other
    float tsum, gsum, wc[49];
    {
        float s0 = 0.f, s1 = 0.f;
        int b = b0 + sl;
        for (; b + 7 * nsl < b1; b += 8 * nsl) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + u * nsl) * ps + row * C0 + c];
            s0 += (v[0] + v[1]) + (v[2] + v[3]); s1 += (v[4] + v[5]) + (v[6] + v[7]);
        }
        for (; b < b1; b += nsl) s0 += part[(size_t)b * ps + row * C0 + c];
        tsum = s0 + s1;
    }
    {   // (the lower-left quadrant of G is the upper-right one transposed)
        const int k2 = tid & 63, s4 = tid >> 6;
        const bool tr = row >= 32 && k2 < 32;
        const size_t go = (size_t)64 * C0 + (tr ? k2 * 64 + row : row * 64 + k2);
        float s0 = 0.f, s1 = 0.f;
        int b = b0 + s4;
        for (; b + 28 < b1; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + 4 * u) * ps + go];
            s0 += (v[0] + v[1]) + (v[2] + v[3]); s1 += (v[4] + v[5]) + (v[6] + v[7]);
        }
        for (; b < b1; b += 4) s0 += part[(size_t)b * ps + go];
        gsum = s0 + s1;
    }
#pragma unroll
    for (int t2 = 0; t2 < 49; ++t2) wc[t2] = w[t2 * C0 + c];
    bn_prologue(bn, 2, C0, 0, C0, cst, C0, blockIdx.x == 0 && blockIdx.y == 0, 256);
    red[tid] = tsum; gred[tid] = gsum;
    __syncthreads();
    if (tid < 64) grow[tid] = (gred[tid] + gred[64 + tid]) + (gred[128 + tid] + gred[192 + tid]);
    __syncthreads();
    if (tid < C0) {
        float T = 0.f;
        for (int k = 0; k < nsl; ++k) T += red[k * C0 + c];
        float gw = 0.f;
#pragma unroll
        for (int t2 = 0; t2 < 49; ++t2) gw = fmaf(grow[(t2 / 7) * 8 + (t2 % 7)], wc[t2], gw);
        const float scale = cst[c], mean = cst[2 * C0 + c], rstd = cst[3 * C0 + c], c2 = cst[4 * C0 + c], c3 = cst[5 * C0 + c];
        const float xb = (bias[c] - mean) * rstd, Cc = -scale * c3;
        const float A = scale, E = Cc * rstd, F = -scale * c2 + Cc * xb;
        atomicAdd(dw + t * C0 + c, (A * T + E * gw + grow[7] * F) * bn.ginv);
    }
}

// dw[(ky*7 + kx)][c] += sum over blocks of part[block][ky*8 + kx][c]; grid (49 taps, WS_SLICES slices of the blocks)
__global__ __launch_bounds__(256) void stem16_wsum_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dw, int C0, int* __restrict__ counter, float ginv) {
    __shared__ float red[256];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *counter = 0;
    const int t = blockIdx.x, row = (t / 7) * 8 + (t % 7);
    const int c = threadIdx.x % C0, sl = threadIdx.x / C0, nsl = 256 / C0;
    const int per = (nblk + WS_SLICES - 1) / WS_SLICES, b0 = blockIdx.y * per, b1 = min(b0 + per, nblk);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0 + sl;
    for (; b + 3 * nsl < b1; b += 4 * nsl) {                                         // four loads in flight
        s0 += part[((size_t)b * 64 + row) * C0 + c]; s1 += part[((size_t)(b + nsl) * 64 + row) * C0 + c];
        s2 += part[((size_t)(b + 2 * nsl) * 64 + row) * C0 + c]; s3 += part[((size_t)(b + 3 * nsl) * 64 + row) * C0 + c];
    }
    for (; b < b1; b += nsl) s0 += part[((size_t)b * 64 + row) * C0 + c];
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < C0) {
        float a = 0.f;
        for (int k = 0; k < nsl; ++k) a += red[k * C0 + c];
        atomicAdd(dw + t * C0 + c, a * ginv);                                            // ginv: the gradient scale of the fp16 build leaves here
    }
}

// ---------------------------------------------------------------------------------------------------- host side
static inline unsigned magic(int d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ULL / (unsigned long long)d); }

static bool stem_geo(StemG& g, int B, int T, int D, int Ho, int Wo, int C0, int pt, int pl, int Hp, int Wp, int ppt, int ppl) {
    if (B < 1 || T < 1 || D < 1 || Ho < 1 || Wo < 1 || (C0 != 32 && C0 != 64) || pt < 0 || pl < 0 || pt > 6 || pl > 6) return false;
    if (Hp < 1 || Wp < 1 || ppt < 0 || ppt > 2 || ppl < 0 || ppl > 2) return false;
    if ((long)B * Ho * Wo * 2 >= (1L << 31) || (long)B * T * D >= (1L << 31)) return false;
    memset(&g, 0, sizeof(g));
    g.B = B; g.T = T; g.D = D; g.Ho = Ho; g.Wo = Wo; g.C0 = C0; g.pt = pt; g.pl = pl; g.Hp = Hp; g.Wp = Wp; g.ppt = ppt; g.ppl = ppl;
    g.pld = 2 * (Wo - 1) + 8;
    g.mg_wo = magic(Wo); g.mg_wp = magic(Wp);
    g.fast = (D % 2 == 0 && pl % 2 == 0 && g.pld <= 128) ? 1 : 0;
    return true;
}
static void stem_tiles(StemG& g, int TR, int rows, int conv_rows_per_tile) {
    g.TR = TR; g.tiles = (rows + TR - 1) / TR;
    const int plane = (((2 * (conv_rows_per_tile - 1) + 8) * g.pld * 2) + 15) & ~15;
    g.ppe = plane / 2; g.patch_bytes = (X3 ? 3 : 1) * plane;
}
template <typename KT>
static bool allow_lds(KT kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
constexpr int WG_MAX_BLOCKS = 512;

}  // namespace

extern "C" {

typedef BnP sarBnRef;

// Forward pass 1: sum / sum of squares of y = conv7x7/2(x) + bias per channel, added into bins [sar_bn_bins()][2][C0] (zeroed by the
// caller once per step, like every BatchNorm of the bf16 trunk).  x (B,T,D) fp32, w (7,7,1,C0) fp32 HWIO, C0 in {32, 64}.
int SAR_STEM(sar_stem16_stats)(const float* x, const float* w, const float* bias, float* bins, int B, int T, int D, int Ho, int Wo, int C0,
                     int pad_top, int pad_left, int det, hipStream_t stream) {
    StemG g;
    if (!x || !w || !bias || !bins || !stem_geo(g, B, T, D, Ho, Wo, C0, pad_top, pad_left, 1, 1, 0, 0)) return SAR_EINVAL;
    int TR = env_int("SAR_STEM_STATS_ROWS", 8); TR = TR < 1 ? 1 : (TR > Ho ? Ho : TR);
    stem_tiles(g, TR, Ho, TR);
    const size_t lds = (size_t)NPL * (C0 / 32) * 4096 + g.patch_bytes + 8 * 2 * C0 * 4;
    if (lds > 160 * 1024) return SAR_EINVAL;
    const int nblk = B * g.tiles;
    if (C0 == 64) {
        if (!allow_lds(stem16_stats_kernel<2>, lds)) return SAR_ELAUNCH;
        hipLaunchKernelGGL(stem16_stats_kernel<2>, dim3(nblk), dim3(256), lds, stream, x, w, bias, bins, g, det);
    } else {
        if (!allow_lds(stem16_stats_kernel<1>, lds)) return SAR_ELAUNCH;
        hipLaunchKernelGGL(stem16_stats_kernel<1>, dim3(nblk), dim3(256), lds, stream, x, w, bias, bins, g, det);
    }
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// Forward pass 2: pooled (B,Hp,Wp,C0) bf16 = maxpool3x3/2(relu(bn(conv(x) + bias))), idx (same shape, u8) = window slot kh*3+kw of the
// first maximum.  bn: statistics from bn->bins (training; the launch with bn->write also stores saved / moving statistics) or
// bn->saved (bins null: inference with folded moving averages).
// (sar_stem16_pool_fwd_x3: pooled = a plane pair, pooled_h -- nullable -- the same map as one fp16 tensor)
#if SAR_X3
int sar_stem16_pool_fwd_x3(const float* x, const float* w, const float* bias, const sarBnRef* bn, void* pooled, void* pooled_h, uint8_t* idx, int B, int T, int D,
                           int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp, int pool_pad_top, int pool_pad_left, hipStream_t stream) {
#else
int SAR_STEM(sar_stem16_pool_fwd)(const float* x, const float* w, const float* bias, const sarBnRef* bn, void* pooled, uint8_t* idx, int B, int T, int D,
                        int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp, int pool_pad_top, int pool_pad_left, hipStream_t stream) {
    void* pooled_h = nullptr;
#endif
    StemG g;
    if (!x || !w || !bias || !bn || !pooled || !idx || !stem_geo(g, B, T, D, Ho, Wo, C0, pad_top, pad_left, Hp, Wp, pool_pad_top, pool_pad_left)) return SAR_EINVAL;
    int TR = env_int("SAR_STEM_POOL_ROWS", 3); TR = TR < 1 ? 1 : (TR > Hp ? Hp : TR);
    const int crows = 2 * TR + 1;
    stem_tiles(g, TR, Hp, crows);
    const size_t lds = (size_t)4 * C0 * 4 + (size_t)NPL * (C0 / 32) * 4096 + g.patch_bytes + (size_t)crows * Wo * ZP;
    if (lds > 160 * 1024) return SAR_EINVAL;
    const int nblk = B * g.tiles;
    if (C0 == 64) {
        if (!allow_lds(stem16_pool_kernel<2>, lds)) return SAR_ELAUNCH;
        hipLaunchKernelGGL(stem16_pool_kernel<2>, dim3(nblk), dim3(256), lds, stream, x, w, bias, *bn, static_cast<u16*>(pooled), static_cast<u16*>(pooled_h), idx, g);
    } else {
        if (!allow_lds(stem16_pool_kernel<1>, lds)) return SAR_ELAUNCH;
        hipLaunchKernelGGL(stem16_pool_kernel<1>, dim3(nblk), dim3(256), lds, stream, x, w, bias, *bn, static_cast<u16*>(pooled), static_cast<u16*>(pooled_h), idx, g);
    }
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// Backward pass 1: the BatchNorm-backward sums (sum dz | sum dz*xhat) of the stem's BatchNorm, added into bn->bbins, from the
// gradient of the pooled map g (bf16), the pooled map and its argmax slots.  Needs bn->saved (written by the forward).  work: the
// workspace sar_stem16_wgrad will get (its tile counter is reset here).
static int stem16_bwd_reduce_impl(const void* g_pooled, const void* pooled, const uint8_t* idx, const float* x, const float* w, const float* bias,
                                  const sarBnRef* bn, float* work, int B, int T, int D, int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp,
                                  int pool_pad_top, int pool_pad_left, bool ph, hipStream_t stream);
int SAR_STEM(sar_stem16_bwd_reduce)(const void* g_pooled, const void* pooled, const uint8_t* idx, const float* x, const float* w, const float* bias,
                          const sarBnRef* bn, float* work, int B, int T, int D, int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp,
                          int pool_pad_top, int pool_pad_left, hipStream_t stream) {
    return stem16_bwd_reduce_impl(g_pooled, pooled, idx, x, w, bias, bn, work, B, T, D, Ho, Wo, C0, pad_top, pad_left, Hp, Wp, pool_pad_top, pool_pad_left, false, stream);
}
#if SAR_X3
// include/sarnet.h: sar_stem16_bwd_reduce_x3 with `pooled` = the FP16 copy of the pooled map (sar_stem16_pool_fwd_x3: pooled_h)
int sar_stem16_bwd_reduce_x3h(const void* g_pooled, const void* pooled_h, const uint8_t* idx, const float* x, const float* w, const float* bias,
                              const sarBnRef* bn, float* work, int B, int T, int D, int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp,
                              int pool_pad_top, int pool_pad_left, hipStream_t stream) {
    return stem16_bwd_reduce_impl(g_pooled, pooled_h, idx, x, w, bias, bn, work, B, T, D, Ho, Wo, C0, pad_top, pad_left, Hp, Wp, pool_pad_top, pool_pad_left, true, stream);
}
#endif
static int stem16_bwd_reduce_impl(const void* g_pooled, const void* pooled, const uint8_t* idx, const float* x, const float* w, const float* bias,
                                  const sarBnRef* bn, float* work, int B, int T, int D, int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp,
                                  int pool_pad_top, int pool_pad_left, bool ph, hipStream_t stream) {
    StemG g;
    if (!g_pooled || !pooled || !idx || !x || !w || !bias || !bn || !bn->bbins || !bn->saved || !work ||
        !stem_geo(g, B, T, D, Ho, Wo, C0, pad_top, pad_left, Hp, Wp, pool_pad_top, pool_pad_left)) return SAR_EINVAL;
    const long npix = (long)B * Hp * Wp;
    if (npix * C0 >= (1L << 31)) return SAR_EINVAL;
    const int npg = 256 / (C0 / 8);
    int nblk = (int)((npix + npg - 1) / npg); nblk = nblk > 1024 ? 1024 : nblk;
    if (ph && X3)
        hipLaunchKernelGGL(stem16_bwd_reduce_kernel<true>, dim3(nblk), dim3(256), 0, stream, static_cast<const u16*>(g_pooled), static_cast<const u16*>(pooled), idx,
                           x, w, bias, bn->gamma, bn->beta, bn->saved, const_cast<float*>(bn->bbins), reinterpret_cast<int*>(work), g, (unsigned)npix);
    else
        hipLaunchKernelGGL(stem16_bwd_reduce_kernel<false>, dim3(nblk), dim3(256), 0, stream, static_cast<const u16*>(g_pooled), static_cast<const u16*>(pooled), idx,
                           x, w, bias, bn->gamma, bn->beta, bn->saved, const_cast<float*>(bn->bbins), reinterpret_cast<int*>(work), g, (unsigned)npix);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

#if !SAR_HALF && !SAR_X3
size_t sar_stem16_wgrad_workspace_bytes(int C0) { return 64 + (size_t)WG_MAX_BLOCKS * (64 * C0 + 64 * 64) * sizeof(float); }   // (round 6: + every block's share of the patches' Gram matrix)
#endif

// Backward pass 2: dw (7,7,1,C0) fp32 += kernel gradient.  bn: saved + bbins (after sar_stem16_bwd_reduce); the launch owns the
// BatchNorm's parameter gradients when bn->write.  work: sar_stem16_wgrad_workspace_bytes(C0) bytes of scratch, the same buffer
// sar_stem16_bwd_reduce got.  The stem's bias
// gradient is exactly zero (it feeds a BatchNorm) and is not computed.
int SAR_STEM(sar_stem16_wgrad)(const float* x, const float* w, const float* bias, const void* g_pooled, const void* pooled, const uint8_t* idx,
                     const sarBnRef* bn, float* dw, float* work, int B, int T, int D, int Ho, int Wo, int C0, int pad_top, int pad_left,
                     int Hp, int Wp, int pool_pad_top, int pool_pad_left, hipStream_t stream) {
    StemG g;
    if (!x || !w || !bias || !g_pooled || !pooled || !idx || !bn || !bn->bbins || !bn->saved || !dw || !work ||
        !stem_geo(g, B, T, D, Ho, Wo, C0, pad_top, pad_left, Hp, Wp, pool_pad_top, pool_pad_left)) return SAR_EINVAL;
    // tile height: 8 conv rows (4 / 5 / 6 / 8 / 10 measured within 0.4 % of each other, round 3); split precision: 6, so that two blocks still fit a
    // CU's LDS (three patch planes, two sets of kernel fragments)
    int TR = env_int("SAR_STEM_WG_ROWS", X3 ? 6 : 8); TR = TR < 1 ? 1 : (TR > Ho ? Ho : TR);
    stem_tiles(g, TR, Ho, TR);
    const int NT = C0 / 32;
    const int dz_bytes = (TR * Wo * DZP + 15) & ~15;                  // + 256 bytes behind it: the scatter's per-lane dummy words
    size_t lds = (size_t)9 * C0 * 4 + (size_t)NPL * NT * 4096 + g.patch_bytes + dz_bytes + 256 + 4 * 3 * PLANE;
    const size_t merge = (size_t)9 * C0 * 4 + 64 * C0 * 4;
    if (lds < merge) lds = merge;
    if (lds > 160 * 1024) return SAR_EINVAL;
    const int ntiles = B * g.tiles;
    const int nblk = ntiles < WG_MAX_BLOCKS ? ntiles : WG_MAX_BLOCKS;
    int* counter = reinterpret_cast<int*>(work);
    float* part = work + 16;
    // The second form (round 6: stem16_wgrad2_kernel -- no conv recomputation, the dense part of dy through the patches' Gram matrix) where the patch
    // fill has its fast form (D and the left padding even, a patch row of at most 128 elements: every configuration of the reference's 80-dim
    // features); SAR_STEM_WG2=0 keeps the first form, for A/B runs.
    static const int wg2 = env_int("SAR_STEM_WG2", 1);
    if (wg2 && g.fast) {
        int TR2 = env_int("SAR_STEM_WG_ROWS", 8); TR2 = TR2 < 1 ? 1 : (TR2 > Ho ? Ho : TR2);
        if (2 * (TR2 - 1) + 8 > 24) TR2 = 9;                               // (the fast fill holds 24 patch rows)
        StemG g2 = g;
        stem_tiles(g2, TR2, Ho, TR2);
        g2.patch_bytes = (g2.patch_bytes / (X3 ? 3 : 1) + 15) & ~15;        // ONE plane of the weight gradient's 2-byte type
        const int DZ2 = C0 * 2 + 8;
        const int dz2 = ((TR2 * Wo + 31) / 32 * 32 * DZ2 + 15) & ~15;
        size_t lds2 = (size_t)g2.patch_bytes + dz2 + 256 + 4 * 2 * PLANE;
        const size_t merge2 = (size_t)(64 * C0 + 64 * 64) * 4;
        if (lds2 < merge2) lds2 = merge2;
        if (lds2 <= 160 * 1024) {
            const int ntiles2 = B * g2.tiles, nblk2 = ntiles2 < WG_MAX_BLOCKS ? ntiles2 : WG_MAX_BLOCKS;
            if (C0 == 64) {
                if (!allow_lds(stem16_wgrad2_kernel<2>, lds2)) return SAR_ELAUNCH;
                hipLaunchKernelGGL(stem16_wgrad2_kernel<2>, dim3(nblk2), dim3(256), lds2, stream, x, static_cast<const u16*>(g_pooled), static_cast<const u16*>(pooled), idx,
                                   counter, part, g2, ntiles2, dz2);
            } else {
                if (!allow_lds(stem16_wgrad2_kernel<1>, lds2)) return SAR_ELAUNCH;
                hipLaunchKernelGGL(stem16_wgrad2_kernel<1>, dim3(nblk2), dim3(256), lds2, stream, x, static_cast<const u16*>(g_pooled), static_cast<const u16*>(pooled), idx,
                                   counter, part, g2, ntiles2, dz2);
            }
            hipLaunchKernelGGL(stem16_wfin_kernel, dim3(49, WS_SLICES), dim3(256), 0, stream, part, nblk2, w, bias, *bn, dw, C0, counter);
            SAR_CHECK_LAUNCH();
            return SAR_OK;
        }
    }
    // SAR_X3 (round 5): g_pooled is the fp16 tensor the mixed backward's last input-gradient launches wrote, times the static scale 1 / bn->ginv,
    // exactly as in the fp16 build: the BatchNorm sums carry the scale, bn->ginv undoes it in dgamma / dbeta and (stem16_wsum) in dw
    sarBnRef bnk = *bn;
    const float gs = 1.f;
    if (C0 == 64) {
        if (!allow_lds(stem16_wgrad_kernel<2>, lds)) return SAR_ELAUNCH;
        hipLaunchKernelGGL(stem16_wgrad_kernel<2>, dim3(nblk), dim3(256), lds, stream, x, w, bias, static_cast<const u16*>(g_pooled),
                           static_cast<const u16*>(pooled), idx, bnk, counter, part, g, ntiles, dz_bytes, gs);
    } else {
        if (!allow_lds(stem16_wgrad_kernel<1>, lds)) return SAR_ELAUNCH;
        hipLaunchKernelGGL(stem16_wgrad_kernel<1>, dim3(nblk), dim3(256), lds, stream, x, w, bias, static_cast<const u16*>(g_pooled),
                           static_cast<const u16*>(pooled), idx, bnk, counter, part, g, ntiles, dz_bytes, gs);
    }
    hipLaunchKernelGGL(stem16_wsum_kernel, dim3(49, WS_SLICES), dim3(256), 0, stream, part, nblk, dw, C0, counter, bn->ginv);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

#ifdef SAR_TIMING
int SAR_STEM(sar_dbg_set_timing_buffer_stem)(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tbuf_stem), &buf, sizeof(buf)) == hipSuccess ? SAR_OK : SAR_ELAUNCH;
}
#endif

}  // extern "C"
