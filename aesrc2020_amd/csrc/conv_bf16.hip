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
#include <hip/hip_runtime.h>
// In-kernel phase timing (tools/conv16_timing.py; built only with -DSAR_TIMING into tools/libsarnet_timing16.so, never part
// of libsarnet_hip.so): thread 0 of every block stamps s_memtime (shader clock) / s_memrealtime (100 MHz, device-wide).
// 32 slots per block: 0-15 the phases, 16-31 the sub-phases of the prologue (trunk16.h: bn_finish stamps through the same macro).
#ifdef SAR_TIMING
namespace {
__device__ unsigned long long* g_tbuf16 = nullptr;
}
#define TSTAMP(slot)                                                                                                   \
    do {                                                                                                               \
        if (g_tbuf16 && threadIdx.x == 0) {                                                                            \
            const long bl = (long)blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);          \
            g_tbuf16[bl * 32 + (slot)] = ((slot) == 0 || (slot) == 15) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
        }                                                                                                              \
    } while (0)
#else
#define TSTAMP(slot) do {} while (0)
#endif
#include "trunk16.h"
#include <stdlib.h>
#include <string.h>

namespace {



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


// row table entry of padded row Rp: {element offset of the source row's first pixel | NONE, owner output pixel of that pixel}
__device__ __forceinline__ uint2 row_entry(const Geo& g, unsigned Rp) {
    const unsigned b = udiv(Rp, g.mg_hp, (unsigned)g.Hp), hp = Rp - b * g.Hp;
    const int dm = g.d - 1, u = (int)hp - g.pt, us = u >> dm;
    const bool ok = (u >= 0) & ((u & dm) == 0) & (us < g.Hs) & (b < (unsigned)g.B);
    uint2 rt;
    rt.x = ok ? ((b * g.Hs + us) * g.Ws) * g.Cin : NONE;
    rt.y = b * (unsigned)(g.Ho * g.Wo) + (dm ? 2u * us : (unsigned)us >> (g.s - 1)) * g.Wo;
    return rt;
}
// byte offset, inside a patch that starts at padded row Rp_lo and holds nrows rows, of the first tap of output pixel `pix` (row stride rs)
__device__ __forceinline__ unsigned patch_pos(const Geo& g, unsigned pix, unsigned Rp_lo, unsigned nrows, int rs) {
    unsigned Rp, wq;
    pix_rp(g, pix, Rp, wq);
    unsigned prow = Rp - Rp_lo;
    if (prow + g.K > nrows) prow = 0;                              // only with the cap_rows clamp
    return (prow * g.Wp + wq) * rs;
}

// Per-thread constants of 8 consecutive channels (a 16-byte piece), read once from the LDS table.
struct Cst8 { float sc[8], sh[8], mu[8], rs[8], c2[8], c3[8]; };
__device__ __forceinline__ void load_cst8(const float* cst, int cs, int ch, int mode, Cst8& k) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        k.sc[j] = cst[ch + j]; k.sh[j] = cst[cs + ch + j];
        if (mode == 2) { k.mu[j] = cst[2 * cs + ch + j]; k.rs[j] = cst[3 * cs + ch + j]; k.c2[j] = cst[4 * cs + ch + j]; k.c3[j] = cst[5 * cs + ch + j]; }
    }
}
// One 16-byte piece (eight channels) of a tensor: `h` from its (hi) plane; SAR_X3: `l` from the lo plane as well.  (Plain uint4 values,
// never a struct of them: arrays of structs written in unrolled loops went to scratch.)
__device__ __forceinline__ void pc_values(const uint4 h, const uint4 l, float (&f)[8]) {
    if constexpr (X3) f32x8_of(h, l, f); else unpack8(h, f);     // (SAR_X3: h, l = the two 16-byte pieces of eight fp32 elements)
}
// MODE 0: copy; 1: relu(x*scale+shift); 2: BatchNorm(+ReLU) backward  dx = scale*(dz - c2 - xhat*c3), dz = da*[z>0];
// 3: mode 2 + a third tensor (the shortcut gradient of the residual block behind this BatchNorm).  fp32 in, fp32 out.
template <int MODE>
__device__ __forceinline__ void xform(const uint4 qh, const uint4 ql, const uint4 q2h, const uint4 q2l, const uint4 q3h, const uint4 q3l, const Cst8& k, float (&o)[8]) {
    float a[8];
    pc_values(qh, ql, a);
    if constexpr (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = a[j];
    } else if constexpr (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaxf(fmaf(a[j], k.sc[j], k.sh[j]), 0.f);
    } else {
        float y[8];
        pc_values(q2h, q2l, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = fmaf(y[j], k.sc[j], k.sh[j]);
            const float dz = z > 0.f ? a[j] : 0.f;
            const float xh = (y[j] - k.mu[j]) * k.rs[j];
            o[j] = k.sc[j] * (dz - k.c2[j] - xh * k.c3[j]);
        }
        if constexpr (MODE == 3) {
            float ad[8];
            pc_values(q3h, q3l, ad);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += ad[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------- conv (fwd / dgrad)
struct ConvP {
    // (the planned instantiations, PL = true: what their requests at kernel entry need sits in ONE 64-byte piece at the front of the argument block
    // -- a round of kernel-argument loads is a scalar-cache round trip every wave pays before its first request can leave, and the compiler loads
    // the fields in the order of their use.  Measured on one box, r6: 1.966 ms per step with the copies and the scheduling barrier behind the entry
    // requests, 1.974 without; the unplanned instantiations read the originals: 1.984 vs 2.006 with the copies)
    const uint4* pl_thr;        // plan tables, below; null: every block derives them itself (the instantiations with PL = false)
    const unsigned* pl_tile;
    const unsigned* pl_lane;
    const float* ent_bins; const float* ent_bbins;   // fb.bins / fb.bbins (copies: bn_issue's addresses)
    int pl_ts, cap_rows_e, fill_e, tmv_e, ent_det, rpi;   // (cap_rows, fill, tmv, fb.det: copies)
    Geo g; int Cout;
    const u16* x; const u16* x2; const u16* x3; const h16x8* wpk; const float* bias; const u16* res; u16* y;
    u16* side;                  // nullable: the fill transform T(x) written back to HBM (same shape as x) for the weight gradient
                                // (SAR_X3: as ONE fp16 tensor, fp16(T(x) * side_scale))
    u16* side2;                 // SAR_X3 only, nullable: T(x) in the trunk's own format (the materialised gradient of fill 3, which further convs read)
    float side_scale;
                                // (SAR_X3: x, x2, x3, res, y, e_y, side2 are fp32 tensors -- trunk16.h -- behind the 2-byte pointer types of the other builds)
    int lplane;                 // SAR_X3: byte offset of the lo plane of the LDS patch
    int tmv;                    // output pixels a block OWNS (= its tile height, except the 32-pixel BR form on 16: SAR_X3 sources whose 32-pixel patch
                                // does not fit the LDS -- the MFMA rows past tmv recompute the next block's pixels and are dropped)
    int fill; BnP fb;
    int epi; float* e_bins; const u16* e_y; const float* e_saved;
    int lg_pc;                  // log2(Cin/8): 16-byte pieces per position
    int rs;                     // LDS row stride in bytes = Cin*2 + 16 (conflict-free ds_read_b128 across rows)
    int Fs;                     // fragment stride of the weight stream per 32 output channels: NPL*K*K*Cin/16 + 8 (zero tail)
    int cap_rows;               // patch rows the LDS allocation holds
    int patch_bytes;            // LDS bytes of the patch region (all planes, incl. a 1 KiB scratch slot each); the weight ring follows
    int rd;                     // depth of the weight ring: 3, or 4 where the LDS has room (128+-channel stride-1 layers)
    int cw; unsigned mg_cw;     // 16-byte pieces of a patch row (padded width x Cin / 8) and the magic number of the division by it
    // Plan tables (round 6; null: every block derives them itself, as rounds 2-5 did).  Everything below is a function of the launch's SHAPE --
    // geometry, tile, thread count -- and was recomputed by every wave of every block of every launch: ~2 000 cycles of dependent integer
    // arithmetic (three divisions per thread for the fill's piece, four for the tile's first and last row, the row table, two more per lane for
    // the A address) in front of the first load of the fill.  conv_plan_kernel writes them ONCE per shape into a workspace the caller lends
    // (sar_conv_plan_workspace_*); a block then asks for them with three loads at kernel entry, in front of the statistic bins.
    //   pl_thr  [threads] {coloff, own_col (without - m0), lo_col, rsub | c8 << 16 | ok_col << 31}: the fill's per-thread constants
    //   pl_tile [pixel tiles][pl_ts] {row table rows (uint2) x cap_rows, {nrows, Rp_lo}}
    //   pl_lane [pixel tiles][TM] byte offset of the pixel's patch position (a_addr without the lane half / k-group terms)
    //   pl_ts: words per tile of pl_tile; rpi: patch rows per trip of the fill (NTHR / cw)
};
constexpr int ROWTAB_BYTES = 1024;   // per patch row {source row base offset | NONE, owner pixel of the row's first column}

// asm-issued LDS-DMA (hidden from hipcc's vmcnt bookkeeping: the waits below are counted by hand; guide section 5.7)
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
// the same with a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset: no 64-bit per-lane address to keep alive across the kernel
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// Workgroup barrier WITHOUT the vmcnt(0) drain of __syncthreads().  It has to be an asm statement with a memory clobber: the
// s_barrier builtin is IntrNoMem, so the compiler may move LDS reads from behind it to in front of it (observed once the
// 128-register variants gave the scheduler a reason: ring fragments read before the other waves' DMA had landed).  The caller
// waits for its own LDS stores / DMA first (gfx950 puts no implicit wait in front of s_barrier).
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_vm_n(int n) {      // wave-uniform n in [0, 63]: vmcnt is a 6-bit counter; larger n wait for 63
#define WVM4(b) case b: wait_vm<b>(); break; case b + 1: wait_vm<b + 1>(); break; case b + 2: wait_vm<b + 2>(); break; case b + 3: wait_vm<b + 3>(); break;
    switch (n) {
        WVM4(0) WVM4(4) WVM4(8) WVM4(12) WVM4(16) WVM4(20) WVM4(24) WVM4(28) WVM4(32) WVM4(36) WVM4(40) WVM4(44) WVM4(48) WVM4(52) WVM4(56)
        case 60: wait_vm<60>(); break; case 61: wait_vm<61>(); break; case 62: wait_vm<62>(); break;
        default: wait_vm<63>(); break;
    }
#undef WVM4
}

// 16-byte store that is not kept in the writer's L2 (nobody on this XCD reads it again before the launch ends)
// SAR_ST_WT (experiment switch, 0 in the shipped build): 1 = the conv's output stores, 2 = also the fills' side stores, as sc1 write-through stores --
// the bytes leave the XCD's L2 while the block still works instead of in the end-of-kernel write-back (guide: a boundary costs + dirty bytes / 6 TB/s)
#ifndef SAR_ST_WT
#define SAR_ST_WT 1
#endif
// Columns a thread owns in the epilogue's row pass (round 5: 4 -- one 16-byte read per k-group partial, a 16-byte residual load and ONE output
// store of 16 bytes (fp32 tensors; 8 bytes in the 2-byte builds) per row instead of two 8-byte rounds: half the instructions of the row pass,
// and write-through stores at the width where they cost what a plain store costs)
#ifndef SAR_EPI_EV
#define SAR_EPI_EV 4
#endif
template <int W> __device__ __forceinline__ void ld_words(const void* p, unsigned (&o)[W]) {
    if constexpr (W == 1) o[0] = *reinterpret_cast<const unsigned*>(p);
    else if constexpr (W == 2) { const uint2 t = *reinterpret_cast<const uint2*>(p); o[0] = t.x; o[1] = t.y; }
    else { const uint4 t = *reinterpret_cast<const uint4*>(p); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
}
// the conv's output store: W 32-bit words.  SAR_ST_WT >= 1: sc1 (write-through: the bytes leave this XCD's L2 while the block still works instead
// of in the end-of-kernel write-back -- r5: 2.128 -> 2.106 ms per step with 8-byte stores); else non-temporal
template <int W> __device__ __forceinline__ void st_out(void* dst, const unsigned (&v)[W]) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
#if SAR_ST_WT >= 1
    // (round 6: the 16-byte form carries its own wait states.  A VMEM store of more than 64 bits must not be followed at once by a VALU write of its
    // data registers; hipcc's hazard recognizer pads its OWN stores and cannot see into an asm statement.  That -- not a cache effect -- was round 5's
    // "write-through anomaly" of the Adam sweep: `global_store_dwordx4 v[72:73], v[60:63], off sc1` straight in front of `v_pk_mul_f32 v[60:61], ...`,
    // 12 of 64 lanes of every chunk wrong; tools/sc1_adam_repro.py.  None of this file's 156 such stores happened to be followed by one -- by register
    // allocation, not by construction; tests/test_host_cpu.py now checks the rule on the built code.)
    if constexpr (W == 1) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(dst), "v"(v[0]) : "memory");
    else if constexpr (W == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst), "v"(u32x2_{v[0], v[1]}) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(u32x4_{v[0], v[1], v[2], v[3]}) : "memory");
#else
    if constexpr (W == 1) __builtin_nontemporal_store(v[0], reinterpret_cast<unsigned*>(dst));
    else if constexpr (W == 2) __builtin_nontemporal_store(u32x2_{v[0], v[1]}, reinterpret_cast<u32x2_*>(dst));
    else __builtin_nontemporal_store(u32x4_{v[0], v[1], v[2], v[3]}, reinterpret_cast<u32x4_*>(dst));
#endif
}
__device__ __forceinline__ void nt_store16(u16* dst, const uint4 v) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#if SAR_ST_WT >= 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(u32x4{v.x, v.y, v.z, v.w}) : "memory");
#else
    __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(dst));
#endif
}

// Patch fill.  rowtab[prow] = {element offset of the source row's first pixel (or NONE), owner output pixel of that pixel};
// per 16-byte piece: one table read, the column test, the transform, one LDS store (SAR_X3: one per plane) -- and, for the block that
// owns the source pixel (the output pixel m_own lies in this block's tile, first channel tile only), the store of T(x) to `side`.
// `mid` runs ONCE, behind the issue of the first pieces' loads and in front of their transform: the caller finishes the BatchNorm constants
// there (trunk16.h: bn_finish), so the bins' round trip, this function's address arithmetic and the pieces' round trip overlap.  Every
// thread runs every trip (pieces past the end are branch-free no-ops), so `mid` may hold barriers.
// A thread keeps ONE column piece (padded column wp, channel group c8) and walks the patch rows: the column test, the channel offset and
// the LDS column offset are per-thread constants, a piece costs one row-table read and ~10 instructions.  cw = Wp * pieces per position
// threads share a row, NTHR / cw rows are in flight per trip (the launcher checks cw <= NTHR); the threads beyond run no-op pieces.
// The single-plane builds (bf16, fp16) keep round 3's fill verbatim (fill_patch16 / xform16: the merged form below cost them 13 registers,
// i.e. spills on the 128- and 168-register instantiations); fill_patch below is the SAR_X3 form and dispatches.
// A32 ("activations fp32", the mixed backward of the split-precision mode, round 5): the tensors of the FORWARD pass a gradient launch reads --
// x2, the BatchNorm input of the fill's backward transform, and e_y, the BatchNorm input of the epilogue's backward sums -- are fp32 tensors
// (what sar_conv2d_x3 left in HBM), while everything that is a gradient (x, x3, res, y, side) is this build's 2-byte type: the input-gradient
// chain of the fp32-class mode runs single-pass on fp16 (tools/grad_recipe_emulator.py: flat gradient 9.6e-4 against 6.8e-4 with split-precision
// input gradients -- the weight gradients' fp16 operands set both) at the fp16 build's cost.  q2 / q2b: the two 16-byte pieces of eight fp32.
template <int MODE, bool A32 = false>
__device__ __forceinline__ uint4 xform16(const uint4 q, const uint4 q2, const uint4 q2b, const uint4 q3, const Cst8& k) {
    if constexpr (MODE == 0) return q;
    float a[8], o[8];
    unpack8(q, a);
    if constexpr (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaxf(fmaf(a[j], k.sc[j], k.sh[j]), 0.f);
    } else {
        float y[8];
        if constexpr (A32) f32x8_of(q2, q2b, y); else unpack8(q2, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = fmaf(y[j], k.sc[j], k.sh[j]);
            const float dz = z > 0.f ? a[j] : 0.f;
            const float xh = (y[j] - k.mu[j]) * k.rs[j];
            o[j] = k.sc[j] * (dz - k.c2[j] - xh * k.c3[j]);
        }
        if constexpr (MODE == 3) {
            float ad[8];
            unpack8(q3, ad);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += ad[j];
        }
    }
    return pack8(o);
}

// The fill's per-thread constants: from the plan table (one 16-byte load the caller issued at kernel entry) or derived here.
struct FillThr { unsigned rsub, rpi, coloff, own_col, lo_col, c8; bool ok_col; };
template <int NTHR>
__device__ __forceinline__ FillThr fill_thread(const Geo& g, int lg_pc, int rs, int tid, unsigned cw, unsigned mg_cw, unsigned m0,
                                               bool planned, const uint4 pthr, unsigned prpi) {
    FillThr f;
    const unsigned pc = 1u << lg_pc;
    if (planned) {                                                 // (kernel-uniform)
        f.coloff = pthr.x; f.own_col = pthr.y - m0; f.lo_col = pthr.z;
        f.rsub = pthr.w & 0xffffu; f.c8 = (pthr.w >> 16) & 0x7fffu; f.ok_col = (pthr.w >> 31) != 0; f.rpi = prpi;   // (conv_plan_kernel packs them)
        return f;
    }
    f.rsub = udiv((unsigned)tid, mg_cw, cw); f.rpi = udiv((unsigned)NTHR, mg_cw, cw);      // row of the trip, rows per trip
    const unsigned col = tid - f.rsub * cw, wp = col >> lg_pc;     // column piece, padded column
    f.c8 = col & (pc - 1);
    const int dm = g.d - 1, ssh = g.s - 1;
    const int v = (int)wp - g.pl, vs = v >> dm;
    f.ok_col = (f.rsub < f.rpi) & (v >= 0) & ((v & dm) == 0) & (vs < g.Ws);
    f.coloff = (unsigned)vs * g.Cin + f.c8 * 8; f.own_col = (dm ? 2u * vs : (unsigned)vs >> ssh) - m0; f.lo_col = wp * rs + f.c8 * 16;
    return f;
}

template <int MODE, int TM, int U, int NTHR, bool A32, typename MID>
__device__ __forceinline__ void fill_patch16(const Geo& g, const u16* __restrict__ x, const u16* __restrict__ x2, const u16* __restrict__ x3, u16* __restrict__ side,
                                           int lg_pc, int rs, unsigned char* patch, const float* cst, const uint2* rowtab,
                                           unsigned nrows, unsigned m0, int tid, bool first_ntile, unsigned cw, unsigned mg_cw, unsigned tmv,
                                           bool planned, const uint4 pthr, unsigned prpi, MID&& mid) {
    const unsigned dummy = nrows * g.Wp * rs + (tid & 63) * 16;   // 1 KiB behind the patch (the launcher allocates it)
    const FillThr ft = fill_thread<NTHR>(g, lg_pc, rs, tid, cw, mg_cw, m0, planned, pthr, prpi);
    const unsigned rsub = ft.rsub, rpi = ft.rpi, coloff = ft.coloff, own_col = ft.own_col, lo_col = ft.lo_col, c8 = ft.c8;
    const bool ok_col = ft.ok_col;
    const unsigned rowbytes = g.Wp * rs;
    const bool do_side = side != nullptr && first_ntile;           // block-uniform
    Cst8 k;
    uint4 q[U], q2[U], q2b[U], q3[U]; unsigned off[U], lo[U], own[U];
    auto issue = [&](unsigned r0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const unsigned prow = r0 + rpi * j, pr = min(prow, nrows - 1);
            const uint2 rt = rowtab[pr];
            const bool in = (prow < nrows) & (rsub < rpi), ok = ok_col & (prow < nrows) & (rt.x != NONE);
            off[j] = ok ? rt.x + coloff : NONE;
            own[j] = rt.y + own_col;                               // < TM: this block owns the source pixel
            lo[j] = in ? pr * rowbytes + lo_col : dummy;           // branch-free: pieces past the end land in a scratch slot
            const unsigned o = ok ? off[j] : 0u;
            q[j] = *reinterpret_cast<const uint4*>(x + o);
            if constexpr (MODE >= 2 && A32) { q2[j] = ld16f(x2, o); q2b[j] = ld16f(x2, o + 4); }
            else { if constexpr (MODE >= 2) q2[j] = *reinterpret_cast<const uint4*>(x2 + o); else q2[j] = q[j]; q2b[j] = q2[j]; }
            if constexpr (MODE == 3) q3[j] = *reinterpret_cast<const uint4*>(x3 + o); else q3[j] = q[j];
        }
    };
    auto finish = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint4 r = xform16<MODE, A32>(q[j], q2[j], q2b[j], q3[j], k);
            *reinterpret_cast<uint4*>(patch + lo[j]) = off[j] == NONE ? make_uint4(0, 0, 0, 0) : r;
            if (do_side && off[j] != NONE && own[j] < tmv) nt_store16(side + off[j], r);   // (tmv: the pixels this block OWNS, <= TM)
        }
    };
    TSTAMP(17);
    issue(rsub);                                                   // trip 0, peeled: `mid` between its loads and their use
    TSTAMP(18);
    mid();
    if constexpr (MODE != 0) load_cst8(cst, g.Cin, c8 * 8, MODE >= 2 ? 2 : 1, k);
    TSTAMP(22);
    finish();
    TSTAMP(23);
    for (unsigned base = U * rpi; base < nrows; base += U * rpi) { issue(base + rsub); finish(); }   // U rows in flight per thread
}
template <int MODE, int TM, int U, int NTHR, bool A32, typename MID>
__device__ __forceinline__ void fill_patch(const ConvP& p, unsigned char* patch, const float* cst, const uint2* rowtab,
                                           unsigned nrows, unsigned m0, int tid, bool first_ntile, bool planned, const uint4 pthr, MID&& mid) {
    if constexpr (!X3) {
        fill_patch16<MODE, TM, U, NTHR, A32>(p.g, p.x, p.x2, p.x3, p.side, p.lg_pc, p.rs, patch, cst, rowtab, nrows, m0, tid, first_ntile, (unsigned)p.cw, p.mg_cw, (unsigned)p.tmv,
                                             planned, pthr, (unsigned)p.rpi, mid);
        return;
    }
    const Geo& g = p.g;
    const u16* __restrict__ x = p.x; const u16* __restrict__ x2 = p.x2; const u16* __restrict__ x3 = p.x3;
    u16* __restrict__ side = p.side; u16* __restrict__ side2 = p.side2;
    const int lg_pc = p.lg_pc, rs = p.rs;
    const unsigned dummy = nrows * g.Wp * rs + (tid & 63) * 16;   // 1 KiB behind the patch plane (the launcher allocates it)
    const FillThr ft = fill_thread<NTHR>(g, lg_pc, rs, tid, (unsigned)p.cw, p.mg_cw, m0, planned, pthr, (unsigned)p.rpi);
    const unsigned rsub = ft.rsub, rpi = ft.rpi, coloff = ft.coloff, own_col = ft.own_col, lo_col = ft.lo_col, c8 = ft.c8;
    const bool ok_col = ft.ok_col;
    const unsigned rowbytes = g.Wp * rs;
    const bool do_side = (side != nullptr || (X3 && side2 != nullptr)) && first_ntile;           // block-uniform
    Cst8 k;
    uint4 qh[U], ql[U], q2h[U], q2l[U], q3h[U], q3l[U]; unsigned off[U], lo[U], own[U];
    auto issue = [&](unsigned r0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const unsigned prow = r0 + rpi * j, pr = min(prow, nrows - 1);
            const uint2 rt = rowtab[pr];
            const bool in = (prow < nrows) & (rsub < rpi), ok = ok_col & (prow < nrows) & (rt.x != NONE);
            off[j] = ok ? rt.x + coloff : NONE;
            own[j] = rt.y + own_col;                               // < TM: this block owns the source pixel
            lo[j] = in ? pr * rowbytes + lo_col : dummy;           // branch-free: pieces past the end land in a scratch slot
            const unsigned o = ok ? off[j] : 0u;
            if constexpr (X3) {                                    // fp32 tensors: elements o .. o+3 and o+4 .. o+7
                qh[j] = ld16f(x, o); ql[j] = ld16f(x, o + 4);
                if constexpr (MODE >= 2) { q2h[j] = ld16f(x2, o); q2l[j] = ld16f(x2, o + 4); }
                if constexpr (MODE == 3) { q3h[j] = ld16f(x3, o); q3l[j] = ld16f(x3, o + 4); }
            } else {
                qh[j] = *reinterpret_cast<const uint4*>(x + o);
                if constexpr (MODE >= 2) q2h[j] = *reinterpret_cast<const uint4*>(x2 + o);
                if constexpr (MODE == 3) q3h[j] = *reinterpret_cast<const uint4*>(x3 + o);
            }
        }
    };
    auto finish = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const bool none = off[j] == NONE;
            const uint4 z4 = make_uint4(0, 0, 0, 0);
            const bool sd = do_side && !none && own[j] < (unsigned)p.tmv;
            {
                float o[8];
                uint4 rh, rl = z4;
                if constexpr (MODE == 0) {
                    if constexpr (X3) { pc_values(qh[j], ql[j], o); split8(o, rh, rl); }   // (fp32 in HBM: the LDS planes are split here)
                    else rh = qh[j];
                } else {
                    if constexpr (MODE == 1) xform<MODE>(qh[j], X3 ? ql[j] : z4, z4, z4, z4, z4, k, o);
                    else if constexpr (MODE == 2) xform<MODE>(qh[j], X3 ? ql[j] : z4, q2h[j], X3 ? q2l[j] : z4, z4, z4, k, o);
                    else xform<MODE>(qh[j], X3 ? ql[j] : z4, q2h[j], X3 ? q2l[j] : z4, q3h[j], X3 ? q3l[j] : z4, k, o);
                    if constexpr (X3) split8(o, rh, rl); else rh = pack8(o);
                }
                *reinterpret_cast<uint4*>(patch + lo[j]) = none ? make_uint4(0, 0, 0, 0) : rh;
                if constexpr (X3) *reinterpret_cast<uint4*>(patch + p.lplane + lo[j]) = none ? make_uint4(0, 0, 0, 0) : rl;
                if (sd) {
                    if constexpr (X3) {
                        if (side != nullptr) nt_store16(side + off[j], pack8_half(o, p.side_scale));
                        if (side2 != nullptr) st32f(side2, off[j], o, true);
                    } else nt_store16(side + off[j], rh);
                }
            }
        }
    };
    TSTAMP(17);
    issue(rsub);                                                   // trip 0, peeled: `mid` between its loads and their use
    TSTAMP(18);
    mid();
    if constexpr (MODE != 0) load_cst8(cst, g.Cin, c8 * 8, MODE >= 2 ? 2 : 1, k);
    TSTAMP(22);
    finish();
    TSTAMP(23);
    for (unsigned base = U * rpi; base < nrows; base += U * rpi) { issue(base + rsub); finish(); }   // U rows in flight per thread
}

// WN: waves along the output channels (tile 32*(4/WN) pixels x 32*WN channels); NKC: 16-channel blocks per virtual tap
// (2, 4, 8); VT: virtual taps = K*K x channel parts of 128 (Cin = 256: two parts per tap, 512: four): 1, 2, 4, 9, 18, 36.
// Ring forms (BR = false: up to 64 source channels into 32 output channels, 32 source channels into 64+): the packed fragment stream
// of this block's channels is brought into an LDS ring (3 deep; 4 where the LDS has room) by LDS-DMA in stages of 8 fragments per
// channel tile (one barrier per stage), so all waves share one copy and no VGPR or VALU instruction is spent on the B operand.  A stage
// is WAITED FOR ONE BARRIER BEFORE IT IS READ.  Waiting right in front of the barrier that releases its readers is not enough: with a
// wait that usually had to stall (2-deep ring) single waves read stale fragments about once in 300 launches -- one wave's 32 x 32 tile
// wrong, the other waves of the block right (tools/conv16_race.py).  A read-back of the wave's own pieces between vmcnt and the
// barrier cured it (so a piece that has left the issuing wave's vmcnt is ordered against that wave's LDS reads, not yet against
// every other wave's), at 5 % of the step; one full stage of margin costs nothing once the ring is a stage deeper.
// WMX: waves along the pixels (default 4 / WN: four waves).  WMX = 8 with WN = 1 -- 256-pixel tiles on eight waves -- is the form of the
// 32-channel layers since round 3 where two such blocks fit a CU.
// SAR_X3: the stream interleaves the hi and the lo fragment of every 16-channel block (fragment 2f, 2f + 1), a ring stage is four such
// pairs, and a k-step is three MFMAs: lo x hi, hi x lo, hi x hi.
constexpr int BR_PD = X3 ? 2 : 4;    // BR form: k-steps of weight fragments in flight ahead of the MFMA that consumes them: FOUR fragments either way (r4 v10, per step: SAR_X3, two fragments a k-step, 1 / 2 / 3 / 4 / 6 steps = 2.461 / 2.306 / 2.323 / 2.336 / 2.350 ms; fp16 build 2 / 4 / 8 = 1.944 / 1.873 / 1.881 ms)
// BR ("B in registers", round 3; the 64+-channel sources with 64+ output channels: stages 1-3): the block is still 64 pixels x 64 channels,
// but its waves are 2 channel halves x KG k-groups, a wave owns BOTH 32-pixel halves of its 32 channels (two accumulators) on a 1/KG share
// of every tap's channels (two fragments per virtual tap), and its weight fragments -- which no other wave of the block needs -- come
// straight from the packed stream into registers as a stream that stays BR_PD (8) k-steps ahead of the MFMA that consumes them: the
// first eight requested at kernel entry, one more per k-step (the whole set at entry -- 18 / 36 per wave -- stalled the prologue behind
// 144 KB per CU on the 64 B/clk L2 port).  No weight ring, no DMA, no barrier inside the k-phase, and HALF the LDS reads per MFMA (an A fragment
// feeds one MFMA, a B fragment two; round 2's k-phase was bound by LDS reads 2 : 1 -- 2 KB per MFMA -- and by a barrier per 8 fragments).
// (the body is a device function: conv_bf16_kernel runs it for one convolution, conv_bf16_pair_kernel for TWO independent ones in one launch.
// lin / ntot / nty: this block's linear index among the ntot blocks of ITS convolution, of which nty are channel tiles per pixel tile)
template <int WN, int NKC, int VT, int KG = 1, int WMX = 4 / WN, bool BR = false, bool A32 = false, bool PL = false>
__device__ __forceinline__ void conv_body(const ConvP& p, const unsigned lin, const unsigned ntot, const unsigned nty) {
    static_assert(BR || KG == 1, "k-groups exist in the BR form only");
    static_assert(!(X3 && A32), "A32 (fp32 forward tensors beside 2-byte gradients) is a form of the single-plane builds");
    static_assert(!BR || (WN == 2 && (WMX == 2 || WMX == 1) && NKC / KG == 2), "BR: 64 (or 32: WMX = 1, the sources whose 64-pixel patch does not fit the LDS) x 64 tiles, two fragments per virtual tap and wave");
    constexpr int NG = BR ? WN : WMX * WN, NW = NG * KG, NTHR = 64 * NW, KPF = NKC / KG;   // waves per k-group, waves, threads, fragments per virtual tap and group
    constexpr int WM = WMX, TM = 32 * WM, TN = 32 * WN;
    static_assert(BR || (WN * 8) % NW == 0 || (X3 && WN * 4 < NW), "stage DMA must divide evenly");
    constexpr int EV = SAR_EPI_EV;                                 // columns per thread in the epilogue's row pass
    static_assert(NTHR % (TN / EV) == 0 && TM % (NTHR / (TN / EV)) == 0, "epilogue maps must divide evenly");
    constexpr int CP = TN / EV, RG = NTHR / CP, RPT = TM / RG;     // epilogue: column groups, row groups, rows per thread
    constexpr int KS = VT % 9 == 0 ? 3 : 1, NH = VT / (KS * KS);  // taps per side, channel parts per tap
    // fragments per ring stage and channel tile: 8; 4 in the split-precision forms of the 32-channel sources (CO2 below: a 12 KB ring leaves
    // room for TWO eight-wave blocks, 2 x 77.8 KB, in a CU's LDS; the four-wave forms -- the stride-2 entry of stage 1 and its shortcut -- likewise)
    constexpr bool CO2 = X3 && !BR && NKC == 2;
    constexpr int SF = CO2 ? 4 : 8;
    constexpr int F = VT * NKC * NPL, NST = (F + SF - 1) / SF;     // weight fragments per channel tile (SAR_X3: hi, lo interleaved), ring stages
    constexpr int SJ = WN * SF, PW = (SJ + NW - 1) / NW;           // DMA wave-instructions per stage: block / per wave (SJ < NW: the first SJ waves issue one each)
    const int RD = p.rd;                                           // ring depth, 3 or 4 (launcher: 4 where the LDS has room for it)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* cst = reinterpret_cast<float*>(smem);
    const int cst_bytes = 6 * p.g.Cin * 4;                         // the six constant rows of the fill's BatchNorm
    uint2* rowtab = reinterpret_cast<uint2*>(smem + cst_bytes);
    unsigned char* patch = smem + cst_bytes + ROWTAB_BYTES;
    unsigned char* bring = patch + p.patch_bytes;                  // [RD][WN][8][1 KiB]
    const Geo& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), kg = w / NG, wm = (w % NG) / WN, wn = (w % NG) % WN;
    // XCD-aware tile order (round 3).  Workgroups go to the eight XCDs round-robin by their linear id, each XCD behind its own L2: in
    // dispatch order the channel tiles of one pixel tile -- which fill the SAME patch -- and neighbouring pixel tiles -- which share halo
    // rows -- sat on different L2s, and each fetched those rows from the fabric again (PMC: 17.4 MB per stage-2 launch for 4.4 MB of
    // tensors).  Here XCD c takes the c-th contiguous eighth of the sequence (channel tile fastest, then pixel tile).
    const unsigned chunk8 = ntot >> 3, tord = lin < (chunk8 << 3) ? (lin & 7u) * chunk8 + (lin >> 3) : lin;
    const unsigned bx = tord >> __builtin_ctz(nty), by = tord & (nty - 1);   // Cout / TN: a power of two
    const unsigned TMV = (X3 || BR) ? (unsigned)(PL ? p.tmv_e : p.tmv) : (unsigned)TM;   // pixels this block owns
    const unsigned m0 = bx * TMV;
    const int n0 = by * TN;
    const unsigned M = (unsigned)g.M;
    TSTAMP(0); TSTAMP(1);
    // ---- plan tables (ConvP::pl_thr): requested FIRST -- loads return in order, and the row table is needed before the statistic bins are
    constexpr bool planned = PL;                                   // (a template parameter: the planned kernels carry none of the derivations, nor the arguments they read)
    uint4 pthr = make_uint4(0, 0, 0, 0);
    uint2 prow = make_uint2(NONE, 0);
    unsigned pa0 = 0, pa1 = 0;
    if (planned) {
        // (the tile's row count travels as row cap_rows of its table, through LDS like the rows: as a scalar load of a header its miss sat in
        // front of every other request of the kernel -- scalar loads return out of order, the first use waits for all of them)
        const uint2* tp = reinterpret_cast<const uint2*>(p.pl_tile + (size_t)bx * p.pl_ts);
        pthr = p.pl_thr[tid];
        if (tid <= p.cap_rows_e) prow = tp[tid];
        const unsigned* lp = p.pl_lane + (size_t)bx * (32 * WMX) + wm * 32 + (lane & 31);
        pa0 = lp[0];
        if constexpr (BR && WMX == 2) pa1 = lp[32];
    }
    // ---- BatchNorm constants of the fill by the WHOLE block (trunk16.h).  The bins were written by the producer's atomics -- behind the fabric,
    // never in this XCD's L2: ~2 800 cycles -- and the fill cannot transform a piece without them, so they are the FIRST thing this kernel
    // requests (round 4: in-kernel stamps of the split-precision stage-2 block showed them requested 3 500 cycles after entry, behind the
    // epilogue operands and the weight fragments, and the block waiting for them in the fill: prologue 9 000 of 30 600 cycles).  They are
    // summed behind the row table (ring forms) or inside the fill, behind the issue of its first pieces' loads (BR forms: 256 registers to
    // hold them in; on the 128- / 168-register forms the pieces held across it spill).
    BnRaw braw;
    constexpr int CIN = NKC < 8 ? 16 * NKC : 128 * NH;             // source channels of this instantiation (the launcher's table)
    if constexpr (PL) {
        if (p.fill_e) bn_issue<NTHR, CIN>(p.ent_bins, p.ent_bbins, p.ent_det, p.fill_e >= 2 ? 2 : 1, braw);
        __builtin_amdgcn_sched_barrier(0);                         // (nothing of the rest of the kernel in front of these requests)
    } else if (p.fill) bn_issue<NTHR, CIN>(p.fb, p.fill >= 2 ? 2 : 1, braw);

    // ---- epilogue operands that do not depend on the accumulators (bias, residual, BatchNorm constants and input of the
    // backward sums): requested first, consumed ~10 us later
    const int cp = tid % CP, rg = tid / CP;
    const int c = n0 + EV * cp;
    // (the per-channel constants travel with the per-pixel operands in load_epi: held from kernel entry, the eight-wave forms -- 128 registers --
    // spilled them behind a vmcnt(0) at the top of the kernel)
    float bv[EV], scv[EV], shv[EV], muv[EV], rsv[EV];
#pragma unroll
    for (int j = 0; j < EV; ++j) { bv[j] = 0.f; scv[j] = 0.f; shv[j] = 0.f; muv[j] = 0.f; rsv[j] = 0.f; }
    // (a 128-register variant for four blocks per CU -- these loads after the k-phase, two fill pieces in flight -- needs the
    // 32-channel layers' LDS at 40 KB, i.e. a 2-deep weight ring: see the ring comment; with the 3-deep ring they fit three per CU
    // and the variant was slower, 23 vs 17 us per stage-0 launch)
    constexpr bool EARLY = NG != 8;                                 // (the eight-wave form lives on 128 registers: operands loaded behind the k-phase)
    constexpr int WR = X3 ? EV : EV / 2;                            // 32-bit words of the EV residual values of a row (fp32 tensors / 2-byte pairs)
    constexpr int WY = (X3 || A32) ? EV : EV / 2;                   // ... of the EV values of e_y (A32: the forward pass's fp32 tensor)
    unsigned rv[RPT][WR], yv[RPT][WY];
    auto load_epi = [&]() __attribute__((always_inline)) {
        if (p.bias) {
            unsigned t[EV]; ld_words<EV>(p.bias + c, t);
#pragma unroll
            for (int j = 0; j < EV; ++j) bv[j] = __uint_as_float(t[j]);
        }
        if (p.epi == 2) {
            const float* sv = p.e_saved; const int C = p.Cout;
            unsigned t0[EV], t1[EV], t2[EV], t3[EV];
            ld_words<EV>(sv + c, t0); ld_words<EV>(sv + C + c, t1); ld_words<EV>(sv + 2 * C + c, t2); ld_words<EV>(sv + 3 * C + c, t3);
#pragma unroll
            for (int j = 0; j < EV; ++j) { muv[j] = __uint_as_float(t0[j]); rsv[j] = __uint_as_float(t1[j]); scv[j] = __uint_as_float(t2[j]); shv[j] = __uint_as_float(t3[j]); }
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const unsigned m = min(m0 + rg + RG * i, M - 1);
            const size_t o = (size_t)m * p.Cout + c;
#pragma unroll
            for (int j = 0; j < WR; ++j) rv[i][j] = 0u;
#pragma unroll
            for (int j = 0; j < WY; ++j) yv[i][j] = 0u;
            if (p.res) { if constexpr (X3) ld_words<WR>(reinterpret_cast<const float*>(p.res) + o, rv[i]); else ld_words<WR>(p.res + o, rv[i]); }
            if (p.epi == 2) { if constexpr (X3 || A32) ld_words<WY>(reinterpret_cast<const float*>(p.e_y) + o, yv[i]); else ld_words<WY>(p.e_y + o, yv[i]); }
        }
    };
    // (when they are requested: consumed behind the k-phase, ~8 000 cycles of work, so nothing is gained by asking first -- and their
    // address arithmetic, 2 x RPT x planes loads, was the first 2 500 cycles of every block.  BR forms: behind the fill; ring forms, whose
    // hand-counted vmcnt waits in the k-phase would otherwise wait for them: behind the weight requests, in front of the row table.)
    TSTAMP(11);
    // ---- weight ring: stages 0 and 1 on their way before anything else
    const h16x8* wsrc = p.wpk + (size_t)(by * WN) * p.Fs * 64;    // (wave-uniform; the lane's 16 bytes are the DMA's offset operand)
    const unsigned ring0 = lds_addr(bring);
    auto issue_stage = [&](int st, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int j = w + NW * i, nt = j / SF, f8 = j % SF;
            if (SJ % NW == 0 || j < SJ)                            // (wave-uniform)
                dma16s(wsrc + ((size_t)nt * p.Fs + st * SF + f8) * 64, (unsigned)lane * 16u, ring0 + (unsigned)((slot * SJ + j) * 1024));
        }
    };
    constexpr int NF = BR ? VT * KPF : 1, NF0 = NF < BR_PD ? NF : BR_PD; // BR: this wave's k-steps; those whose weight fragments are requested ahead of the k-phase
    h16x8 breg[NF][NPL];
    const h16x8* wmine = p.wpk + ((size_t)(by * WN + wn) * p.Fs + kg * KPF * NPL) * 64 + lane;
    auto load_b = [&](int i) __attribute__((always_inline)) {      // k-step i: fragment (i / KPF) * NKC + i % KPF of this group's share, every plane
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) breg[i][pl] = wmine[(((i / KPF) * NKC + (i % KPF)) * NPL + pl) * 64];
    };
    if constexpr (BR) {
#pragma unroll
        for (int i = 0; i < NF0; ++i) load_b(i);
    } else {
        issue_stage(0, 0);
        if (NST > 1) issue_stage(1, 1);
        if (RD > 3 && NST > 2) issue_stage(2, 2);
    }
    if constexpr (EARLY && !BR) load_epi();
    TSTAMP(12);

    unsigned Rp_lo, nrows;
    if (planned) {
        Rp_lo = 0; nrows = 0;                                      // (nrows: read back from the table behind the barrier; Rp_lo: only a_addr's unplanned arm)
        if (tid <= p.cap_rows) rowtab[tid] = prow;                 // row table: as conv_plan_kernel wrote it (rows past the tile's last: NONE); [cap_rows].x = nrows
    } else {
        unsigned Rp_hi, wtmp;
        pix_rp(g, m0, Rp_lo, wtmp);
        pix_rp(g, min(m0 + TMV - 1, M - 1), Rp_hi, wtmp);
        Rp_hi += g.K - 1;
        nrows = Rp_hi - Rp_lo + 1;
        if (nrows > (unsigned)p.cap_rows) nrows = p.cap_rows;      // never taken: the launcher sizes cap_rows as an upper bound
        if ((unsigned)tid < nrows) rowtab[tid] = row_entry(g, Rp_lo + tid);   // row table
    }
    TSTAMP(13);
    constexpr bool MIDFILL = BR;
    if constexpr (!MIDFILL) {
        if (p.fill) bn_finish<NTHR, CIN>(p.fb, p.fill >= 2 ? 2 : 1, cst, g.Cin, bx == 0 && by == 0, braw, reinterpret_cast<double*>(patch));
    }
    TSTAMP(14);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // raw barrier (no vmcnt drain): row table (and constants) complete
    wg_barrier();
    if (planned) nrows = rowtab[p.cap_rows].x;
    TSTAMP(16);
#ifdef SAR_TIMING_PROBE
    TSTAMP(8);
#endif
    if constexpr (!MIDFILL) TSTAMP(2);
    auto mid = [&]() __attribute__((always_inline)) {
        if constexpr (MIDFILL) {
            if (p.fill) {
                bn_finish<NTHR, CIN>(p.fb, p.fill >= 2 ? 2 : 1, cst, g.Cin, bx == 0 && by == 0, braw, reinterpret_cast<double*>(patch));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wg_barrier();
            }
            TSTAMP(2);
        }
    };

    // (Issuing the patch loads in front of the BatchNorm prologue -- row table, barrier, loads, prologue, barrier, transform + store --
    // was built and measured: no gain.  What the fill costs is its address arithmetic, ~1 400 cycles before the first load leaves,
    // not the round trip; and holding the pieces across the prologue puts the narrow layers past 170 registers.)
    // fill pieces per thread in flight (eight waves, two blocks per CU: 128 registers).  (Eight in the BR forms, to cover the fill's loads in
    // one round trip: 2.626 vs 2.594 ms per split-precision step, same box, 300-step runs -- the registers cost more than the round trip.)
    constexpr int FU = (NG == 8 || X3) ? 2 : 4;                    // (SAR_X3: two 32-byte pieces are what four 16-byte ones were; 2.254 vs 2.259 ms)
    // ---- fill: the padded patch, all input channels, transformed once
    // (A32: a piece of the gradient fills is three / four 16-byte loads -- two pieces in flight)
    if (p.fill == 0) fill_patch<0, TM, FU, NTHR, false>(p, patch, cst, rowtab, nrows, m0, tid, by == 0, planned, pthr, mid);
    else if (p.fill == 1) fill_patch<1, TM, FU, NTHR, false>(p, patch, cst, rowtab, nrows, m0, tid, by == 0, planned, pthr, mid);
#ifndef SAR_A32_FU2
#define SAR_A32_FU2 2
#endif
    else if (p.fill == 2) fill_patch<2, TM, (X3 && FU > 2 ? 2 : (A32 && FU > SAR_A32_FU2) ? SAR_A32_FU2 : FU), NTHR, A32>(p, patch, cst, rowtab, nrows, m0, tid, by == 0, planned, pthr, mid);
    else fill_patch<3, TM, (X3 ? 1 : FU < 2 ? FU : 2), NTHR, A32>(p, patch, cst, rowtab, nrows, m0, tid, by == 0, planned, pthr, mid);   // three tensors: two pieces in flight (SAR_X3: six loads a piece)
    if constexpr (EARLY && BR) load_epi();
    // ---- per-lane A address
    auto a_addr = [&](unsigned pix0, unsigned planned_pos) __attribute__((always_inline)) {
        const unsigned pos = planned ? planned_pos : patch_pos(g, min(pix0 + (lane & 31), M - 1), Rp_lo, nrows, p.rs);
        return pos + (lane >> 5) * 16 + kg * (KPF * 32);           // + the group's channel half of every tap
    };
    const unsigned a_lane = a_addr(m0 + wm * 32, pa0);
    TSTAMP(3);

    // ---- k-phase, straight-line.
    // two accumulators: an instruction between two MFMAs on the SAME accumulator costs ~43 cycles, on different ones ~6
    // (MI355X_MICROARCH.md, per-instruction constants) -- and every MFMA here has LDS reads next to it
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    if constexpr (BR) {
        // acc: pixels 0-31 of the tile, acc2: pixels 32-63 (WMX = 2); both on this wave's 32 channels and its share of every tap's channels.
        // WMX = 1 (32-pixel tiles): acc2 takes the small terms of SAR_X3 (and nothing otherwise).
        unsigned a_hi = 0;
        if constexpr (WMX == 2) a_hi = a_addr(m0 + 32, pa1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this wave's patch stores
        wg_barrier();
        TSTAMP(4);
        auto a_off = [&](int i) __attribute__((always_inline)) {
            const int v = i / KPF, j = i % KPF, tap = v / NH, hh = v % NH, dy = tap / KS, dx = tap % KS;
            return (unsigned)((dy * g.Wp + dx) * p.rs + hh * (NKC * 32) + j * 32);
        };
        h16x8 a0[2][NPL], a1[2][NPL];
        auto lda = [&](int i) __attribute__((always_inline)) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                a0[i & 1][pl] = *reinterpret_cast<const h16x8*>(patch + pl * p.lplane + a_lane + a_off(i));
                if constexpr (WMX == 2) a1[i & 1][pl] = *reinterpret_cast<const h16x8*>(patch + pl * p.lplane + a_hi + a_off(i));
            }
        };
        lda(0);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
#ifndef SAR_TIMING_PROBE
            if (i == NF / 2) TSTAMP(8);
#endif
            if (i + NF0 < NF) load_b(i + NF0);                     // the stream stays NF0 k-steps ahead
            if (i + 1 < NF) lda(i + 1);
            if constexpr (!X3) {
                acc = MFMA16(a0[i & 1][0], breg[i][0], acc, 0, 0, 0);
                if constexpr (WMX == 2) acc2 = MFMA16(a1[i & 1][0], breg[i][0], acc2, 0, 0, 0);
            } else if constexpr (WMX == 2) {                       // small terms first; the two accumulators alternate
                acc = MFMA16(a0[i & 1][NPL - 1], breg[i][0], acc, 0, 0, 0);
                acc2 = MFMA16(a1[i & 1][NPL - 1], breg[i][0], acc2, 0, 0, 0);
                acc = MFMA16(a0[i & 1][0], breg[i][NPL - 1], acc, 0, 0, 0);
                acc2 = MFMA16(a1[i & 1][0], breg[i][NPL - 1], acc2, 0, 0, 0);
                acc = MFMA16(a0[i & 1][0], breg[i][0], acc, 0, 0, 0);
                acc2 = MFMA16(a1[i & 1][0], breg[i][0], acc2, 0, 0, 0);
            } else {
                acc2 = MFMA16(a0[i & 1][NPL - 1], breg[i][0], acc2, 0, 0, 0);
                acc = MFMA16(a0[i & 1][0], breg[i][0], acc, 0, 0, 0);
                acc2 = MFMA16(a0[i & 1][0], breg[i][NPL - 1], acc2, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                     // (the scheduler otherwise sinks the stream's loads to their uses: distance 1-2 instead of NF0)
        }
    } else {
        // Per stage: wait for its fragments, barrier (also: every wave is done with the ring slot refilled next; on stage 0: the patch is
        // complete), start the DMA of stage + 2, read the stage's 8 B fragments, then its MFMAs; the A fragments of the NEXT virtual tap
        // are read from the patch while the current tap's MFMAs issue.
        // (SAR_X3: the stage's fragments are read a k-step -- one hi / lo pair -- at a time, one k-step ahead: 16 registers instead of 32,
        // which is what lets the eight-wave 32-channel form live on 128 registers, two blocks per CU)
        h16x8 afr[2][NKC][NPL], bfr[X3 ? 4 : 8];
        const unsigned char* bmine = bring + (wn * SF) * 1024 + lane * 16;
        auto load_afrags = [&](int s, unsigned addr) __attribute__((always_inline)) {
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) afr[s][kc][pl] = *reinterpret_cast<const h16x8*>(patch + pl * p.lplane + addr + kc * 32);
        };
#pragma unroll
        for (int v = 0; v < VT; ++v) {
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const int f = (v * NKC + kc) * NPL;                // first fragment of this k-step in the stream
                if (f % SF == 0) {
                    const int st = f / SF;
                    // stages <= st+1 have landed -- stage st one whole stage and a barrier ago (see above); RD = 4: stage st+2 in flight
                    if (RD == 4 && st + 2 < NST) wait_vm<PW>(); else wait_vm<0>();      // (kernel-uniform branch)
                    // stage 0: this wave's patch stores must have completed before the others read them (gfx950 inserts no
                    // implicit wait in front of a bare s_barrier)
                    if (st == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    wg_barrier();
                    if (st == 0) { TSTAMP(4); load_afrags(0, a_lane); }
#ifndef SAR_TIMING_PROBE
                    if (st == NST / 2) TSTAMP(8);
#endif
                    if (RD == 4) { if (st + 3 < NST) issue_stage(st + 3, (st + 3) & 3); }
                    else if (st + 2 < NST) issue_stage(st + 2, (st + 2) % 3);
                    const int slot = RD == 4 ? (st & 3) : (st % 3);
                    if constexpr (!X3) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) bfr[i] = *reinterpret_cast<const h16x8*>(bmine + (slot * SJ + i) * 1024);
                    } else {
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) bfr[((f / NPL) & 1) * NPL + pl] = *reinterpret_cast<const h16x8*>(bmine + (slot * SJ + pl) * 1024);
                    }
                }
                if constexpr (X3) {                                // the next k-step's pair, while it is in the same stage
                    if ((f + NPL) % SF != 0 && f + NPL < F) {
                        const int st = f / SF, slot = RD == 4 ? (st & 3) : (st % 3);
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl)
                            bfr[((f / NPL + 1) & 1) * NPL + pl] = *reinterpret_cast<const h16x8*>(bmine + (slot * SJ + (f + NPL) % SF + pl) * 1024);
                    }
                }
                if (kc == 0 && v + 1 < VT) {
                    const int tap = (v + 1) / NH, hh = (v + 1) % NH, dy = tap / KS, dx = tap % KS;
                    load_afrags((v + 1) & 1, a_lane + (unsigned)((dy * g.Wp + dx) * p.rs + hh * (NKC * 32)));
                }
                if constexpr (!X3) {
                    if (f & 1) acc2 = MFMA16(afr[v & 1][kc][0], bfr[f % 8], acc2, 0, 0, 0);
                    else acc = MFMA16(afr[v & 1][kc][0], bfr[f % 8], acc, 0, 0, 0);
                } else {
                    constexpr int NP1 = NPL - 1;
                    const int bq = ((f / NPL) & 1) * NPL;
                    acc2 = MFMA16(afr[v & 1][kc][NP1], bfr[bq], acc2, 0, 0, 0);
                    acc = MFMA16(afr[v & 1][kc][0], bfr[bq], acc, 0, 0, 0);
                    acc2 = MFMA16(afr[v & 1][kc][0], bfr[bq + NP1], acc2, 0, 0, 0);
                }
            }
        }
    }
    if constexpr (!(BR && WMX == 2)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    }
    TSTAMP(5);
    if constexpr (!EARLY) load_epi();
    lds_barrier();                                                 // the patch is dead: reuse it as the output staging tile
    TSTAMP(10);

    // ---- epilogue: accumulators -> LDS (fp32) -> row-major pass: + bias (+ residual), 2-byte stores, statistics
    float* ot = reinterpret_cast<float*>(patch) + kg * (TM * TN);  // [KG][TM][TN]: each k-group stages its partial tile
    float* part = reinterpret_cast<float*>(patch) + KG * TM * TN;  // [RG][TN][2]
    {
        const int col = wn * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            ot[row * TN + col] = acc[r];
            if constexpr (BR && WMX == 2) ot[(row + 32) * TN + col] = acc2[r];
        }
    }
    lds_barrier();
    TSTAMP(9);
    float s1[EV], s2[EV];
    long long q1[EV], q2[EV];                                      // epi 3: the same sums in 64-bit fixed point (trunk16.h: fix28)
#pragma unroll
    for (int j = 0; j < EV; ++j) { s1[j] = 0.f; s2[j] = 0.f; q1[j] = 0; q2[j] = 0; }
    typedef __attribute__((ext_vector_type(EV))) float fEV;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int row = rg + RG * i;
        const unsigned m = m0 + row;
        fEV a = *reinterpret_cast<const fEV*>(reinterpret_cast<const float*>(patch) + row * TN + EV * cp);
#pragma unroll
        for (int kk = 1; kk < KG; ++kk)                             // the other k-groups' partial tiles
            a += *reinterpret_cast<const fEV*>(reinterpret_cast<const float*>(patch) + kk * (TM * TN) + row * TN + EV * cp);
        float v[EV];
#pragma unroll
        for (int j = 0; j < EV; ++j) {
            v[j] = a[j] + bv[j];
            if constexpr (X3) v[j] += __uint_as_float(rv[i][j]);
            else v[j] += (j & 1) ? h_hi(rv[i][j >> 1]) : h_lo(rv[i][j >> 1]);
        }
        if (m < M && (unsigned)row < TMV) {
            if constexpr (X3) {                                    // the fp32 values as they are
                unsigned w[EV];
#pragma unroll
                for (int j = 0; j < EV; ++j) w[j] = __float_as_uint(v[j]);
                st_out<EV>(reinterpret_cast<float*>(p.y) + (size_t)m * p.Cout + c, w);
            } else {
                unsigned w[EV / 2];
#pragma unroll
                for (int j = 0; j < EV / 2; ++j) w[j] = pack2(v[2 * j], v[2 * j + 1]);
                st_out<EV / 2>(p.y + (size_t)m * p.Cout + c, w);
                // forward statistics: of what was stored (the consumer normalises the stored values)
                if (p.epi == 1 || p.epi == 3) {
#pragma unroll
                    for (int j = 0; j < EV / 2; ++j) { v[2 * j] = h_lo(w[j]); v[2 * j + 1] = h_hi(w[j]); }
                }
            }
            // backward sums: of the fp32 values (they are parameter gradients -- near-cancelling sums that the 2^-9 storage rounding would dominate)
            if (p.epi == 1) {
#pragma unroll
                for (int j = 0; j < EV; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
            } else if (p.epi == 3) {
#pragma unroll
                for (int j = 0; j < EV; ++j) { q1[j] += fix28(v[j]); q2[j] += fix28(fminf(v[j] * v[j], 8.0e6f)); }
            } else if (p.epi == 2) {
#pragma unroll
                for (int j = 0; j < EV; ++j) {
                    float y;
                    if constexpr (X3 || A32) y = __uint_as_float(yv[i][j]);
                    else y = (j & 1) ? h_hi(yv[i][j >> 1]) : h_lo(yv[i][j >> 1]);
                    const float d = fmaf(y, scv[j], shv[j]) > 0.f ? v[j] : 0.f;
                    s1[j] += d; s2[j] += d * (y - muv[j]) * rsv[j];
                }
            }
        }
    }
    TSTAMP(6);
    if (p.epi == 0) { TSTAMP(7); TSTAMP(15); return; }              // kernel-uniform
    if (p.epi == 3) {                                              // integer partials, integer atomics: order-independent
        long long* partq = reinterpret_cast<long long*>(part);
#pragma unroll
        for (int j = 0; j < EV; ++j) { partq[(rg * TN + EV * cp + j) * 2 + 0] = q1[j]; partq[(rg * TN + EV * cp + j) * 2 + 1] = q2[j]; }
        lds_barrier();
        if (tid < 2 * TN) {
            const int cc = tid >> 1, which = tid & 1;
            long long s = 0;
#pragma unroll
            for (int r = 0; r < RG; ++r) s += partq[(r * TN + cc) * 2 + which];
            atomicAdd(reinterpret_cast<unsigned long long*>(p.e_bins) + ((bx % NB) * 2 + which) * p.Cout + n0 + cc, (unsigned long long)s);
        }
        TSTAMP(7); TSTAMP(15);
        return;
    }
#pragma unroll
    for (int j = 0; j < EV; ++j) { part[(rg * TN + EV * cp + j) * 2 + 0] = s1[j]; part[(rg * TN + EV * cp + j) * 2 + 1] = s2[j]; }
    lds_barrier();
    if (tid < 2 * TN) {
        const int cc = tid >> 1, which = tid & 1;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < RG; ++r) s += part[(r * TN + cc) * 2 + which];
        atomicAdd(p.e_bins + ((bx % NB) * 2 + which) * p.Cout + n0 + cc, s);
    }
    TSTAMP(7); TSTAMP(15);
}

// ---- plan tables of one launch shape (ConvP::pl_thr / pl_tile / pl_lane), written once per shape by the launcher
struct PlanArgs {
    Geo g; int lg_pc, rs, cw; unsigned mg_cw; int nthr, tmv, TM, cap_rows, ntiles, pl_ts;
    uint4* thr; unsigned* tile; unsigned* lane;
};
__global__ __launch_bounds__(256) void conv_plan_kernel(const PlanArgs a) {
    const Geo& g = a.g;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x == a.ntiles) {                              // the fill's per-thread constants (fill_thread with m0 = 0)
        for (int t = tid; t < a.nthr; t += 256) {
            const unsigned cw = (unsigned)a.cw, pc = 1u << a.lg_pc;
            const unsigned rsub = udiv((unsigned)t, a.mg_cw, cw), rpi = udiv((unsigned)a.nthr, a.mg_cw, cw);
            const unsigned col = t - rsub * cw, wp = col >> a.lg_pc, c8 = col & (pc - 1);
            const int dm = g.d - 1, ssh = g.s - 1;
            const int v = (int)wp - g.pl, vs = v >> dm;
            const bool ok_col = (rsub < rpi) & (v >= 0) & ((v & dm) == 0) & (vs < g.Ws);
            a.thr[t] = make_uint4((unsigned)vs * g.Cin + c8 * 8, dm ? 2u * vs : (unsigned)vs >> ssh, wp * a.rs + c8 * 16,
                                  rsub | (c8 << 16) | (ok_col ? 0x80000000u : 0u));
        }
        return;
    }
    const unsigned bx = blockIdx.x, M = (unsigned)g.M, m0 = bx * (unsigned)a.tmv;
    unsigned Rp_lo, Rp_hi, wtmp;
    pix_rp(g, min(m0, M - 1), Rp_lo, wtmp);
    pix_rp(g, min(m0 + (unsigned)a.tmv - 1, M - 1), Rp_hi, wtmp);
    Rp_hi += g.K - 1;
    unsigned nrows = Rp_hi - Rp_lo + 1;
    if (nrows > (unsigned)a.cap_rows) nrows = a.cap_rows;
    unsigned* tp = a.tile + (size_t)bx * a.pl_ts;                   // rows 0 .. cap_rows-1, then {nrows, Rp_lo}
    if (tid == 0) { tp[2 * a.cap_rows] = nrows; tp[2 * a.cap_rows + 1] = Rp_lo; }
    for (int t = tid; t < a.cap_rows; t += 256) {
        const uint2 rt = (unsigned)t < nrows ? row_entry(g, Rp_lo + t) : make_uint2(NONE, 0u);
        tp[2 * t] = rt.x; tp[2 * t + 1] = rt.y;
    }
    for (int i = tid; i < a.TM; i += 256) a.lane[(size_t)bx * a.TM + i] = patch_pos(g, min(m0 + i, M - 1), Rp_lo, nrows, a.rs);
}

// waves per SIMD the register budget is set for.  BR: 1 (KG = 4: 512 threads, one block per CU) or 2; ring forms: 3 (two waves), 4 (eight
// waves: 128 registers, two blocks per CU), 2 otherwise.  SAR_X3 doubles the fragments a wave holds and its LDS admits one eight-wave block
// (or two four-wave blocks) per CU anyway: 2 for every ring form.
#define CONV_OCC (BR ? (KG == 4 ? 1 : 2) : X3 ? (WMX * WN == 8 && NKC == 2 ? 4 : 2) : (WMX * WN == 2 ? 3 : WMX * WN == 8 ? 4 : 2))

template <int WN, int NKC, int VT, int KG = 1, int WMX = 4 / WN, bool BR = false, bool A32 = false, bool PL = false>
__global__ __launch_bounds__(64 * (BR ? WN : WMX * WN) * KG, CONV_OCC) void conv_bf16_kernel(const ConvP p) {
    conv_body<WN, NKC, VT, KG, WMX, BR, A32, PL>(p, blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y, gridDim.y);
}

// TWO independent convolutions of equal block size in ONE launch (round 3): the 1x1 shortcut of a residual block that changes the shape
// (resnet.py:82-87) next to the block's first 3x3 conv -- same input, different outputs -- forward, and their input gradients backward.
// A launch inside the step costs ~4.5 us before it does anything, the shortcut launches are 7-20 us of mostly that, and the
// 3x3 launches of the stages they sit in front of leave CUs idle.  Blocks [0, na) run the first convolution, the rest the second (1x1:
// same WN / NKC / KG / WMX / BR, VT = channel parts of ONE tap); LDS and registers are the larger of the two.
// (WMXB, round 6: the second convolution's pixel-tile class where it differs -- the stride-2 entry of the last stage runs 32-pixel tiles because its
// patch does not fit otherwise, its 1x1 shortcut 64-pixel ones; BR forms only: the block size does not depend on WMX there.)
template <int WN, int NKC, int VT, int KG, int WMX, bool BR, bool A32 = false, bool PL = false, int WMXB = WMX>
__global__ __launch_bounds__(64 * (BR ? WN : WMX * WN) * KG, CONV_OCC)
void conv_bf16_pair_kernel(const ConvP pa, const ConvP pb, const unsigned na, const unsigned ntya, const unsigned nb, const unsigned ntyb) {
    const unsigned na8 = (na + 7u) & ~7u;                             // the second convolution starts on XCD 0 as well (its XCD-aware tile order)
    if (blockIdx.x < na) conv_body<WN, NKC, VT, KG, WMX, BR, A32, PL>(pa, blockIdx.x, na, ntya);
    else if (blockIdx.x >= na8) conv_body<WN, NKC, (VT % 9 == 0 ? VT / 9 : VT), KG, WMXB, BR, A32, PL>(pb, blockIdx.x - na8, nb, ntyb);
}

// ---------------------------------------------------------------------------------------------------- weight gradient
#if !SAR_X3   // (the SAR_X3 mode runs the fp16 build's weight-gradient kernels on fp16 operands: trunk16.h)
// dw[tap][ci][co] += sum over pixels of A[pixel + tap][ci] * dY[pixel][co] on PLAIN bf16 operands: A = the activated tensor the
// forward conv's fill wrote back (ConvP::side; the raw tensor for shortcuts / the first conv), dY = the gradient tensor (for
// conv1 of a block: the BatchNorm-backward the input-gradient conv's fill wrote back).  Both arrive by LDS-DMA in 64-pixel
// chunks and feed the MFMA through ds_read_b64_tr_b16 (hardware transpose); no VALU work on the data at all.
// ONE launch computes the weight gradients of MANY convolutions (they are off the critical path of the backward pass, so the
// trunk runs them at the end of each backward part): a block takes one work item = (convolution, channel tile, a run of pixel
// chunks) from a table the host planned (sar_wgrad_plan_bf16: longest items first) and adds its tiles into the gradient --
// fp32 atomics when the convolution's pixels are split over several items, plain read-add-write otherwise.
struct WgConv {                 // device table row, written by sar_wgrad_plan_bf16 (128 bytes)
    Geo g; int Cout;
    const u16* x; const u16* dy; float* dw; float* dbias;
    int nchunks;                // chunks of CH pixels (64; 128 for the 3x3 launches since round 6)
    int cap_pos;                // patch positions of one chunk (multiple of 16)
    short la;                   // 3x3: chunks requested ahead of the one being multiplied (1..3: what the position ring holds)
    short ch;                   // 3x3: pixels per chunk of this convolution, 64 or 128 (nchunks, cap_pos, la and the items' runs count in it)
    float ginv;                 // the flush multiplies by it: 1 / (gradient scale dy carries; 1 in the bf16 build)
};
static_assert(sizeof(WgConv) <= 128, "WgConv must fit its 128-byte table row");
struct WgItem { int conv, ci0, co0, c_first, c_count, single, slab, pad1_; };    // 32 bytes; slab: partial-tile slot of a non-single item
// Flush of an item whose tile is shared with other items (its convolution's pixels are split over several runs): the raw partial
// tile goes to the item's own slab of a workspace -- plain 16-byte stores in accumulator-register order (1 KiB per wave-instruction) --
// and wgrad_reduce_kernel, launched behind, sums a tile's slabs in a fixed order and adds the result into the gradient.  Round 2 added
// every partial tile with fp32 atomics: 40 % of the 64-wide launch (147 KB of device-scope atomics per item, executed behind the
// fabric) and an order-dependent sum.
struct WgTile { int conv, ci0, co0, first, count, pad0_, pad1_, pad2_; };          // 32 bytes: the slices of one gradient tile: slices[first .. first+count)
__device__ uint4 g_zero_page[64];
constexpr int WG_NBUF = 4;      // 1x1: chunk buffers per wave (three chunks in flight behind the one being multiplied)
constexpr int WG_TAB_MAX = 6144; // 3x3: positions in an item's source-offset table (24 KB of LDS at most)
constexpr int WG_NW1 = 4;       // 1x1: independent waves per block, each on its own share of the item's chunks (merged in LDS at the end)


// 1x1 convolutions (the shortcuts, resnet.py:82-87): the "patch" of a chunk is its 64 pixels (strided gather for stride 2).  A
// block's four waves split the item's chunks and never synchronise inside the loop: each wave runs its own DMA ring (8 KB per
// chunk: dY [64][32] | x [64][32]) and waits on its own vmcnt; the four 32 x 32 tiles are merged in LDS at the end.
__global__ __launch_bounds__(WG_NW1 * 64) void wgrad1_bf16_kernel(const unsigned char* __restrict__ convs, const WgItem* __restrict__ items,
                                                                 const uint4* __restrict__ zero, float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr unsigned BUF = 8192;
    const WgItem it = items[blockIdx.x];
    const WgConv* cvp = reinterpret_cast<const WgConv*>(convs + (size_t)it.conv * 128);
    const Geo g = cvp->g;                                                          // by value: scalar registers for the whole kernel
    const int Cout = cvp->Cout;
    const u16* __restrict__ xp = cvp->x; const u16* __restrict__ dyp = cvp->dy;
    float* dwp = cvp->dw; float* dbp = cvp->dbias;
    const float ginv = cvp->ginv;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = it.ci0, co0 = it.co0;
    const unsigned M = (unsigned)g.M;
    const int cw = (it.c_count + WG_NW1 - 1) / WG_NW1;
    const int c_first = it.c_first + wv * cw, nmine = max(0, min(cw, it.c_count - wv * cw));
    unsigned char* bufs = smem + (size_t)wv * WG_NBUF * BUF;
    const unsigned bufs0 = lds_addr(bufs);
    const unsigned row = lane >> 2, pc8 = (lane & 3) * 8;                          // a DMA instruction = 16 pixels x 32 channels
    TSTAMP(0); TSTAMP(1); TSTAMP(2);
    auto issue = [&](int lc) __attribute__((always_inline)) {                      // 8 instructions: 4 for dY, 4 for x
        const unsigned mc0 = (unsigned)(c_first + lc) * 64;
        const unsigned dst0 = bufs0 + (unsigned)(lc % WG_NBUF) * BUF;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned m = mc0 + (j & 3) * 16 + row;
            const void* src;
            if (j < 4) {
                src = m < M ? (const void*)(dyp + (size_t)m * Cout + co0 + pc8) : (const void*)(zero + lane);
            } else {
                unsigned Rp, wq;
                pix_rp(g, min(m, M - 1), Rp, wq);
                const unsigned off = src_off(g, Rp, wq, m < M);
                src = off != NONE ? (const void*)(xp + off + ci0 + pc8) : (const void*)(zero + lane);
            }
            dma16(src, dst0 + (unsigned)j * 1024);
        }
    };
    for (int k = 0; k < WG_NBUF - 1 && k < nmine; ++k) issue(k);
    TSTAMP(3);

    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    float bsum = 0.f;
    const bool do_bias = dbp != nullptr && ci0 == 0;
    const int gi = lane & 15, gg = lane >> 4, cb = gg & 1, kh = gg >> 1;
    const unsigned colb = (16 * cb + 4 * (gi & 3)) * 2;
    const unsigned prb = 8 * kh + (gi >> 2);
    for (int lc = 0; lc < nmine; ++lc) {
        // chunk lc has landed once at most min(nmine-1-lc, NBUF-2) younger chunks of this wave are still in flight (same wave
        // issues and reads: its vmcnt orders the two)
        wait_vm_n(8 * min(nmine - 1 - lc, WG_NBUF - 2));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the transpose-reads of chunk lc-1 are done: its slot is free
        if (lc + WG_NBUF - 1 < nmine) issue(lc + WG_NBUF - 1);
        const unsigned char* dyt = bufs + (size_t)(lc % WG_NBUF) * BUF;
        const unsigned char* ap = dyt + 4096;
        h16x8 A[4], Bm[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned o0 = (16 * ks + prb) * 64 + colb, o1 = o0 + 4 * 64;
            const s16x4 b_lo = lds_tr(dyt, o0), b_hi = lds_tr(dyt, o1), a_lo = lds_tr(ap, o0), a_hi = lds_tr(ap, o1);
            A[ks] = __builtin_bit_cast(h16x8, __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7));
            Bm[ks] = __builtin_bit_cast(h16x8, __builtin_shufflevector(b_lo, b_hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        acc = MFMA16(A[0], Bm[0], acc, 0, 0, 0);
        acc2 = MFMA16(A[1], Bm[1], acc2, 0, 0, 0);
        acc = MFMA16(A[2], Bm[2], acc, 0, 0, 0);
        acc2 = MFMA16(A[3], Bm[3], acc2, 0, 0, 0);
        if (do_bias) {                                             // block-uniform
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += (float)Bm[ks][j];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    TSTAMP(4);
    // ---- merge the waves' tiles in LDS (each wave's own first buffer: its reads are done), then one add per element
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float* mine = reinterpret_cast<float*>(bufs);
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
    __syncthreads();
#pragma unroll
    for (int e = tid; e < 1024; e += WG_NW1 * 64) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < WG_NW1; ++k) s += reinterpret_cast<const float*>(smem + (size_t)k * WG_NBUF * BUF)[e];
        float* o = dwp + (size_t)(ci0 + (e >> 5)) * Cout + co0 + (e & 31);
        if (it.single) *o += s * ginv; else slab[(size_t)it.slab * 1024 + e] = s;   // slab: natural [32][32] order, raw
    }
    if (do_bias) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (lane < 32) atomicAdd(dbp + co0 + lane, bsum * ginv);
    }
    TSTAMP(5); TSTAMP(15);
}

// ---- 3x3: rolling patch ring and up to 64 x 64 channel tiles --------------------------------------------------------------------
// What the first batched form (one private 160-position patch per chunk, 32 x 32 tiles) measured (profiles/r2_v8_wgrad_pmc.txt: 7.5 M L2 requests = 960 MB per launch for 300 MB of unique operands, L2 hit
// rate 24 %, 2.7 TB/s through the fabric, 260 us): every 64-pixel chunk fetched its whole padded patch -- 160 positions for 64 pixels,
// the halo rows and the partial rows again with the next chunk -- and every 32-channel tile fetched 64-byte halves of 128+-byte rows.
// Here a block streams its run of pixels through ONE circular buffer of padded positions per 32-channel plane: consecutive chunks
// only fetch the positions that are new (~70 per 64 pixels), position p lives at slot p mod NP, a tap is a constant position
// offset (dy*Wp + dx) added before the wrap.  A block owns TCI x TCO planes of 32 channels (64 x 64 where the layer has them): the
// activation planes are shared by the TCO output planes and vice versa, halving the bytes and the transpose-reads per MFMA.
// Per chunk the block issues 4*TCO DMA instructions for the gradient rows and TCI per new 16-position group, dealt round-robin to
// the nine waves; every wave issues the same count (the surplus repeat the last one), so the hand-counted vmcnt stays three
// scalars: the instruction counts of the chunks in flight.
struct Wg3Args { const unsigned char* convs; const WgItem* items; const uint4* zero; unsigned np; int nbuf; int tab_pos; float* slab; };

// W12 (64 x 64 tiles only): TWELVE waves -- wave = (tap row, 32 x 32 sub-tile), three taps each -- instead of nine (wave = tap, the
// whole tile).  Nine waves sit 3-2-2-2 on the four SIMDs, so a chunk's 144 MFMAs cost 1 536 cycles on the fullest SIMD and the other
// six waves wait for it at the chunk's barrier; twelve sit 3-3-3-3 (1 152 cycles), and the DMA issue of a chunk is dealt to twelve.
// Per wave and 16-pixel step: three A fragments (one activation plane at three consecutive positions) and ONE gradient fragment.
// (Negative: extra LOADER waves that issue all of the block's LDS-DMA while the compute waves only read LDS and multiply -- 64-wide launch
// 125 -> 141 us: what a chunk waits for is the ARRIVAL of ~17 KB of 64 / 128-byte pieces per CU, not instruction issue.)
// CH (round 6): pixels per chunk, 64 or 128.  A chunk is what the block synchronises on -- one barrier, one round of DMA issue, one table of
// pixel positions -- and at 64 pixels a block spent 3 870 cycles per chunk for 1 150 of MFMA on its fullest SIMD (tools/conv16_timing.py wgrad:
// the wait for the slowest of twelve waves 1 490, the DMA issue 720).  128 pixels: half the barriers and tables per pixel, the MFMAs of sixteen
// 16-pixel steps back to back; the ring holds LA + 1 chunks of positions as before, so LA is one less where 64-pixel chunks ran three ahead.
// The chunk size is a property of the CONVOLUTION (WgConv::ch, the planner's choice: 128 where the convolution's positions fit the ring with a chunk
// in flight -- the 13 x 3 images of stage 3 do not: a 128-pixel chunk spans three images and their padding rows): a launch whose plan mixes both
// runs the kernel that holds both bodies (MIX) behind a block-uniform branch.
template <int TCI, int TCO, bool W12, int CH>
__device__ __forceinline__ void wgrad3_body(const Wg3Args& a, const WgItem& it, const WgConv* cvp) {
    static_assert(!W12 || (TCI == 2 && TCO == 2), "twelve waves: three tap rows x the four sub-tiles of a 64 x 64 tile");
    static_assert(CH == 64 || CH == 128, "64- or 128-pixel chunks");
    constexpr int NWV = W12 ? 12 : 9, NTHR = NWV * 64;
    constexpr int DYI = CH / 16;                                                    // DMA instructions per gradient plane and chunk (16 pixels each)
    constexpr int NA = W12 ? 3 : TCI, NBF = W12 ? 1 : TCO, NJ = W12 ? 3 : TCI * TCO;   // A / B fragments per step, MFMAs per step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Geo g = cvp->g;
    const int Cout = cvp->Cout, LA = cvp->la;
    const u16* __restrict__ xp = cvp->x; const u16* __restrict__ dyp = cvp->dy;
    float* dwp = cvp->dw; float* dbp = cvp->dbias;
    const float ginv = cvp->ginv;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = it.ci0, co0 = it.co0;
    const unsigned M = (unsigned)g.M, NP = a.np, mask = a.np - 1;
    const int c_first = it.c_first, nmine = it.c_count, nbuf = LA + 1;           // gradient-row slots: the chunk being multiplied + LA in flight
    unsigned* qring = reinterpret_cast<unsigned*>(smem);                          // [2][CH]: absolute patch position of every pixel
    unsigned* soff = qring + 2 * CH;                                               // [tab_pos]: source element offset of every position of this item (NONE: zero)
    unsigned char* dyr = smem + 8 * CH + (size_t)a.tab_pos * 4;                    // [nbuf][TCO][CH px][64 B]
    unsigned char* pr = dyr + (size_t)nbuf * TCO * (CH * 64);                      // [TCI][NP][64 B]
    const unsigned dyr0 = lds_addr(dyr), pr0a = lds_addr(pr);
    constexpr int MAXNI = CH == 64 ? 4 : 6;

    auto tables = [&](int lc) __attribute__((always_inline)) {
        if (tid < CH) {
            const unsigned mc0 = (unsigned)(c_first + lc) * CH;
            unsigned Rp, wq;
            pix_rp(g, min(mc0 + tid, M - 1), Rp, wq);                              // past the batch: the last pixel's position; its dY row is zero
            qring[(lc & 1) * CH + tid] = Rp * g.Wp + wq;
        }
    };
    unsigned Gn;                                                                   // first 16-position group not requested yet
    {
        unsigned Rp, wq;
        pix_rp(g, (unsigned)c_first * CH, Rp, wq);
        Gn = (Rp * g.Wp + wq) >> 4;
    }
    // Source offsets of all positions this item will ever request, once: position -> (image, row, column) costs two divisions,
    // and doing that inside every DMA instruction (~45 VALU instructions each, nine waves at the same time right behind the
    // barrier) made address generation, not the MFMAs, the longest phase of a chunk.
    const unsigned P0 = Gn * 16;
    {
        unsigned Rpl, wql;
        pix_rp(g, min((unsigned)(c_first + nmine) * CH, M) - 1, Rpl, wql);
        const unsigned npos = min((((Rpl + g.K) * g.Wp + 15) & ~15u) - P0, (unsigned)a.tab_pos);
        for (unsigned i = tid; i < npos; i += NTHR) {
            const unsigned p = P0 + i;
            const unsigned Rp = udiv(p, g.mg_wp, (unsigned)g.Wp), wp = p - Rp * g.Wp;
            const unsigned b = udiv(Rp, g.mg_hp, (unsigned)g.Hp), hp = Rp - b * g.Hp;
            const int u = (int)hp - g.pt, v = (int)wp - g.pl;
            const bool ok = (u >= 0) & (u < g.Hs) & (v >= 0) & (v < g.Ws) & (b < (unsigned)g.B);
            soff[i] = ok ? ((b * g.Hs + u) * g.Ws + v) * g.Cin : NONE;
        }
    }
    // per-lane constants of a DMA piece: 4 lanes per position / pixel, 16 bytes = 8 channels each
    const unsigned l4 = lane >> 2, pc8 = (lane & 3) * 8;
    auto issue = [&](int lc) __attribute__((always_inline)) -> int {
        const unsigned mc0 = (unsigned)(c_first + lc) * CH;
        unsigned Rpl, wql;
        pix_rp(g, min(mc0 + CH - 1, M - 1), Rpl, wql);
        const unsigned Gh = ((Rpl + g.K) * g.Wp + 15) >> 4;                        // every position in front of (last row + K) * Wp
        const int npg = Gh > Gn ? (int)(Gh - Gn) : 0;
        const int total = DYI * TCO + npg * TCI;
        const int NI = (total + NWV - 1) / NWV;
        const unsigned dyslot = dyr0 + (unsigned)(lc % nbuf) * (TCO * CH * 64);
        // (Computing the addresses of a batch first and issuing its DMA instructions back to back -- dma16's memory clobber keeps the
        // next instruction's table read behind it -- was measured: 125 -> 134 us for the 64-wide launch.  Issue order is not the limit.)
        auto addr = [&](int j, const void*& src, unsigned& dst) __attribute__((always_inline)) {
            if (j < DYI * TCO) {                                                   // gradient rows: plane j / DYI, 16 pixels j % DYI
                const int pl = j / DYI, blk = j % DYI;
                const unsigned m = mc0 + blk * 16 + l4;
                src = m < M ? (const void*)(dyp + (size_t)m * Cout + co0 + pl * 32 + pc8) : (const void*)(a.zero + lane);
                dst = dyslot + (unsigned)(pl * (CH * 64) + blk * 1024);
            } else {                                                               // 16 new patch positions of one activation plane
                const int jj = j - DYI * TCO, pl = TCI == 1 ? 0 : (jj & 1);
                const unsigned gi = Gn + (unsigned)(TCI == 1 ? jj : jj >> 1);
                const unsigned so = soff[gi * 16 + l4 - P0];
                src = so != NONE ? (const void*)(xp + so + ci0 + pl * 32 + pc8) : (const void*)(a.zero + lane);
                dst = pr0a + (unsigned)pl * (NP * 64) + ((gi * 16) & mask) * 64;
            }
        };
#pragma unroll
        for (int i = 0; i < MAXNI; ++i)                                            // every wave issues, at most MAXNI each
            if (i < NI) {
                const void* src; unsigned dst;
                addr(min(w + NWV * i, total - 1), src, dst);
                dma16(src, dst);
            }
        if (Gh > Gn) Gn = Gh;
        return NI;
    };
    TSTAMP(0); TSTAMP(1);
#ifdef SAR_TIMING
    unsigned long long t_wait = 0, t_issue = 0, t_mma = 0;
#define TACC(var, expr) do { const unsigned long long t0__ = __builtin_amdgcn_s_memtime(); expr; var += __builtin_amdgcn_s_memtime() - t0__; } while (0)
#else
#define TACC(var, expr) do { expr; } while (0)
#endif
    tables(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wg_barrier();                                                                  // the offset table is complete
    int q1 = 0, q2 = 0, q3 = 0;                                                    // DMA instructions of this wave for chunks lc+1, lc+2, lc+3
    int q0 = nmine > 0 ? issue(0) : 0;
    if (LA > 1 && nmine > 1) q1 = issue(1);
    if (LA > 2 && nmine > 2) q2 = issue(2);
    (void)q0;

    // job k of a wave = one 32 x 32 accumulator: nine waves: (i, j) = (k / TCO, k % TCO) of tap w; twelve: tap 3 * (w / 4) + k of sub-tile
    // (wi, wj) = ((w / 2) % 2, w % 2)
    f32x16 acc[NJ];
#pragma unroll
    for (int k = 0; k < NJ; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float bsum[NBF];
#pragma unroll
    for (int j = 0; j < NBF; ++j) bsum[j] = 0.f;
    const int wi = W12 ? (w >> 1) & 1 : 0, wj = W12 ? w & 1 : 0, wrow = W12 ? w >> 2 : 0;
    const bool do_bias = dbp != nullptr && ci0 == 0 && (W12 ? (wrow == 0 && wi == 0) : w == 0);
    const unsigned toff = W12 ? (unsigned)(wrow * g.Wp) : (unsigned)((w / 3) * g.Wp + (w % 3));   // first tap of this wave, in positions
    const int gi_ = lane & 15, gg = lane >> 4, cb = gg & 1, kh = gg >> 1;
    const unsigned colb = (16 * cb + 4 * (gi_ & 3)) * 2;
    const unsigned prb = 8 * kh + (gi_ >> 2);
    TSTAMP(2);

    for (int lc = 0; lc < nmine; ++lc) {
        // chunk lc has landed once only the instructions of the younger chunks in flight are outstanding
        // (chunk lc+1 too: a full step between "landed" and "read", see conv_bf16_kernel's ring)
        TACC(t_wait, wait_vm_n(q2 + q3); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); wg_barrier());   // everyone's pieces landed; everyone is done with chunk lc-1
        int qn = 0;
        TACC(t_issue, if (lc + LA < nmine) qn = issue(lc + LA));   // dY slot of chunk lc+LA-nbuf <= lc-1; ring positions behind chunk lc
        if (LA == 3) q3 = qn; else if (LA == 2) q2 = qn; else q1 = qn;
        if (lc + 1 < nmine) tables(lc + 1);
        const unsigned* qtab = qring + (lc & 1) * CH;
        const unsigned char* dyt = dyr + (size_t)(lc % nbuf) * (TCO * CH * 64);
        h16x8 A[2][NA], Bm[2][NBF];
        auto rd = [&](int ks, h16x8 (&Ao)[NA], h16x8 (&Bo)[NBF]) __attribute__((always_inline)) {
            const unsigned p0 = 16 * ks + prb, p1 = p0 + 4;
            const unsigned q0p = qtab[p0] + toff, q1p = qtab[p1] + toff;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                // nine waves: plane i at the wave's tap; twelve: plane wi at the three taps of the wave's row
                const unsigned char* pl = pr + (size_t)(W12 ? wi : i) * NP * 64;
                const unsigned a0 = ((q0p + (W12 ? i : 0)) & mask) * 64 + colb, a1 = ((q1p + (W12 ? i : 0)) & mask) * 64 + colb;
                const s16x4 lo = lds_tr(pl, a0), hi = lds_tr(pl, a1);
                Ao[i] = __builtin_bit_cast(h16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < NBF; ++j) {
                const unsigned char* pl = dyt + (W12 ? wj : j) * (CH * 64);
                const s16x4 lo = lds_tr(pl, p0 * 64 + colb), hi = lds_tr(pl, p1 * 64 + colb);
                Bo[j] = __builtin_bit_cast(h16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        };
        rd(0, A[0], Bm[0]);
#pragma unroll
        for (int ks = 0; ks < CH / 16; ++ks) {
            if (ks + 1 < CH / 16) rd(ks + 1, A[(ks + 1) & 1], Bm[(ks + 1) & 1]);
#pragma unroll
            for (int k = 0; k < NJ; ++k)
                acc[k] = MFMA16(A[ks & 1][W12 ? k : k / TCO], Bm[ks & 1][W12 ? 0 : k % TCO], acc[k], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < NBF; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[j] += (float)Bm[ks & 1][j][e];
            }
        }
        q1 = q2; q2 = q3; q3 = 0;                                  // the chunks in flight move up one place
    }
    TSTAMP(3);
#ifdef SAR_TIMING
    if (g_tbuf16 && threadIdx.x == 0) {
        const long bl = blockIdx.x;
        g_tbuf16[bl * 32 + 8] = t_wait; g_tbuf16[bl * 32 + 9] = t_issue; g_tbuf16[bl * 32 + 10] = (unsigned long long)nmine;
    }
#endif
    // ---- flush
    {
        const int col = lane & 31;
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            const int tap = W12 ? 3 * wrow + k : w, i = W12 ? wi : k / TCO, j = W12 ? wj : k % TCO;
            float* o = dwp + ((size_t)tap * g.Cin + ci0 + i * 32) * Cout + co0 + j * 32 + col;
            if (it.single) {
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) old[r] = o[(size_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * Cout];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * Cout] = old[r] + acc[k][r] * ginv;
            } else {
                // slab[item][tap][i][j][q][lane] x float4: the accumulator registers as they are (rows 8q + 4(lane/32) + 0..3, column lane%32)
                float4* sl = reinterpret_cast<float4*>(a.slab) + ((((size_t)it.slab * 9 + tap) * TCI + i) * TCO + j) * 256 + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __builtin_nontemporal_store(f32x4{acc[k][4 * q], acc[k][4 * q + 1], acc[k][4 * q + 2], acc[k][4 * q + 3]},
                                                reinterpret_cast<f32x4*>(sl + q * 64));
            }
        }
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < NBF; ++j) {
                float s = bsum[j];
                s += __shfl_xor(s, 32, 64);
                if (lane < 32) atomicAdd(dbp + co0 + (W12 ? wj : j) * 32 + lane, s * ginv);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TSTAMP(4); TSTAMP(15);
}
template <int TCI, int TCO, bool W12 = false, bool MIX = false>
__global__ __launch_bounds__(W12 ? 768 : 576) void wgrad3_bf16_kernel(const Wg3Args a) {
    const WgItem it = a.items[blockIdx.x];
    const WgConv* cvp = reinterpret_cast<const WgConv*>(a.convs + (size_t)it.conv * 128);
    if constexpr (MIX) {
        if (cvp->ch == 128) { wgrad3_body<TCI, TCO, W12, 128>(a, it, cvp); return; }   // (block-uniform)
    }
    wgrad3_body<TCI, TCO, W12, 64>(a, it, cvp);
}

// Sum of the partial tiles of every multi-slice gradient tile, in slice order (deterministic), added into the gradient: grid
// (tiles, 32 x 32 sub-tiles of a tile = taps x TCI x TCO), 256 threads x float4.  K3: slabs in accumulator-register order (above).
// WR (round 5): the sums are WRITTEN, dw = ..., not added -- a gradient tile has exactly one reduction block, and in the train step dw was zeroed at the step's start: the
// read of the read-modify-write was 21 MB of zeros per step (-4 us).
template <bool K3, bool WR = false>
__device__ __forceinline__ void wgrad_reduce_body(const unsigned char* __restrict__ convs, const WgTile* __restrict__ tiles,
                                                  const int* __restrict__ slices, const float* __restrict__ slab, int tci, int tco, int tile, int sub, int nsub) {
    const WgTile t = tiles[tile];
    const WgConv* cvp = reinterpret_cast<const WgConv*>(convs + (size_t)t.conv * 128);
    const int Cin = cvp->g.Cin, Cout = cvp->Cout;
    const float ginv = cvp->ginv;
    float* dwp = cvp->dw;
    const int tid = threadIdx.x;
    const float4* base = reinterpret_cast<const float4*>(slab) + (size_t)sub * 256 + tid;
    const size_t per = (size_t)nsub * 256;                                           // float4 per slab
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    const int* sl = slices + t.first;
    int s = 0;
    for (; s + 3 < t.count; s += 4) {                                                 // four slabs in flight
        const float4 v0 = base[sl[s] * per], v1 = base[sl[s + 1] * per], v2 = base[sl[s + 2] * per], v3 = base[sl[s + 3] * per];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;   a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;   a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; s < t.count; ++s) { const float4 v0 = base[sl[s] * per]; a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w; }
    const float r[4] = {((a0.x + a1.x) + (a2.x + a3.x)) * ginv, ((a0.y + a1.y) + (a2.y + a3.y)) * ginv,
                        ((a0.z + a1.z) + (a2.z + a3.z)) * ginv, ((a0.w + a1.w) + (a2.w + a3.w)) * ginv};
    if constexpr (K3) {
        const int j = sub % tco, i = (sub / tco) % tci, w = sub / (tco * tci);
        const int q = tid >> 6, lane = tid & 63, col = lane & 31, row0 = 8 * q + 4 * (lane >> 5);
        float* o = dwp + ((size_t)w * Cin + t.ci0 + i * 32 + row0) * Cout + t.co0 + j * 32 + col;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (WR) o[(size_t)k * Cout] = r[k]; else o[(size_t)k * Cout] += r[k]; }
    } else {
        float4* o = reinterpret_cast<float4*>(dwp + (size_t)(t.ci0 + (tid >> 3)) * Cout + t.co0 + (tid & 7) * 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!WR) v = *o;
        v.x += r[0]; v.y += r[1]; v.z += r[2]; v.w += r[3];
        *o = v;
    }
}

template <bool K3>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const unsigned char* __restrict__ convs, const WgTile* __restrict__ tiles,
                                                           const int* __restrict__ slices, const float* __restrict__ slab, int tci, int tco) {
    wgrad_reduce_body<K3>(convs, tiles, slices, slab, tci, tco, blockIdx.x, blockIdx.y, gridDim.y);
}
// The reductions of SEVERAL batched launches (the step's four tile classes) in ONE launch: a launch inside the step costs ~4.5 us before it
// does anything, and three of the four reductions are 10-17 us of it.  Block b belongs to plan i with first[i] <= b < first[i+1].
struct RedPlan { const unsigned char* convs; const WgTile* tiles; const int* slices; const float* slab; int n_tiles, k3, tci, tco; };
struct RedPlans { RedPlan p[4]; int first[5]; };
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const RedPlans P) {
    const int b = blockIdx.x;
    const int i = (b >= P.first[1]) + (b >= P.first[2]) + (b >= P.first[3]);
    const RedPlan& q = P.p[i];
    const int local = b - P.first[i], nsub = (q.k3 & 1) ? 9 * q.tci * q.tco : 1, tile = local / nsub, sub = local - tile * nsub;
    if (q.k3 & 2) {                                                 // (block-uniform) bit 1: written, not added
        if (q.k3 & 1) wgrad_reduce_body<true, true>(q.convs, q.tiles, q.slices, q.slab, q.tci, q.tco, tile, sub, nsub);
        else wgrad_reduce_body<false, true>(q.convs, q.tiles, q.slices, q.slab, q.tci, q.tco, tile, sub, nsub);
    } else if (q.k3 & 1) wgrad_reduce_body<true>(q.convs, q.tiles, q.slices, q.slab, q.tci, q.tco, tile, sub, nsub);
    else wgrad_reduce_body<false>(q.convs, q.tiles, q.slices, q.slab, q.tci, q.tco, tile, sub, nsub);
}

#endif  // !SAR_X3
// ---------------------------------------------------------------------------------------------------- BatchNorm passes
// eight consecutive values of a 2-byte tensor at element e (SAR_X3: hi + lo plane, `ps` elements apart)
__device__ __forceinline__ void load8(const u16* __restrict__ t, long e, long ps, float (&f)[8]) {
    if constexpr (X3) f32x8_of(ld16f(t, e), ld16f(t, e + 4), f);   // (SAR_X3: an fp32 tensor; `ps` is unused)
    else unpack8(*reinterpret_cast<const uint4*>(t + e), f);
}
__device__ __forceinline__ void store8(u16* __restrict__ t, long e, long ps, const float (&f)[8]) {
    if constexpr (X3) st32f(t, e, f);
    else *reinterpret_cast<uint4*>(t + e) = pack8(f);
}
// Residual-stream gradient: out = BN(+ReLU) backward of `dact` through the BatchNorm whose input was x (+ add).
template <typename DT>
__global__ __launch_bounds__(256) void bn_bwd_apply_bf16_kernel(const DT* __restrict__ dact, const u16* __restrict__ x, const BnP bn,
                                                                const u16* __restrict__ add, u16* __restrict__ out, u16* __restrict__ out16, long R, int C, int relu,
                                                                float out_scale) {
    // (SAR_X3: `out` is a plane pair in true units, out16 -- nullable -- the same values times out_scale as ONE fp16 tensor, the operand of the weight gradients)
    __shared__ float cst[6 * CMAX];
    bn_prologue(bn, 2, C, 0, C, cst, CMAX, blockIdx.x == 0, 256);
    __syncthreads();
    const int pc = C >> 3;                                         // power of two
    const long total = R * pc, stride = (long)gridDim.x * 256;     // stride % pc == 0
    const int c8 = threadIdx.x & (pc - 1);
    Cst8 k;
    load_cst8(cst, CMAX, c8 * 8, 2, k);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long e = i * 8;
        float a[8], y[8], o[8], ad[8];
        const long ps = R * C;
        if constexpr (sizeof(DT) == 2) load8(reinterpret_cast<const u16*>(dact), e, ps, a);
        else {
            const float4 f0 = *reinterpret_cast<const float4*>(dact + e), f1 = *reinterpret_cast<const float4*>(dact + e + 4);
            a[0] = f0.x; a[1] = f0.y; a[2] = f0.z; a[3] = f0.w; a[4] = f1.x; a[5] = f1.y; a[6] = f1.z; a[7] = f1.w;
        }
        load8(x, e, ps, y);
        if (add) load8(add, e, ps, ad);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = fmaf(y[j], k.sc[j], k.sh[j]);
            const float dz = (!relu || z > 0.f) ? a[j] : 0.f;
            const float xh = (y[j] - k.mu[j]) * k.rs[j];
            o[j] = (k.sc[j] * (dz - k.c2[j] - xh * k.c3[j]) + (add ? ad[j] : 0.f)) * (X3 ? 1.f : out_scale);
        }
        if (!X3 || out != nullptr) store8(out, e, ps, o);            // (SAR_X3, the mixed backward: only the scaled fp16 tensor)
        if constexpr (X3) { if (out16 != nullptr) *reinterpret_cast<uint4*>(out16 + e) = pack8_half(o, out_scale); }
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
        if constexpr (sizeof(DT) == 2) load8(reinterpret_cast<const u16*>(dact), e, R * C, a);
        else {
            const float4 f0 = *reinterpret_cast<const float4*>(dact + e), f1 = *reinterpret_cast<const float4*>(dact + e + 4);
            a[0] = f0.x; a[1] = f0.y; a[2] = f0.z; a[3] = f0.w; a[4] = f1.x; a[5] = f1.y; a[6] = f1.z; a[7] = f1.w;
        }
        load8(x, e, R * C, y);
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
    bn_prologue(bn, 1, C, 0, C, cst, CMAX, blockIdx.x == 0, 256);
    __syncthreads();
    const int pc = C >> 3;
    const long total = R * pc, stride = (long)gridDim.x * 256;
    const int c8 = threadIdx.x & (pc - 1);
    Cst8 k;
    load_cst8(cst, CMAX, c8 * 8, 1, k);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long e = i * 8;
        float y[8];
        load8(x, e, R * C, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) { y[j] = fmaf(y[j], k.sc[j], k.sh[j]); if (relu) y[j] = fmaxf(y[j], 0.f); }
        *reinterpret_cast<float4*>(out + e) = make_float4(y[0], y[1], y[2], y[3]);
        *reinterpret_cast<float4*>(out + e + 4) = make_float4(y[4], y[5], y[6], y[7]);
    }
}

// (SAR_X3: the 2-byte side is a plane pair, n8 * 8 elements apart; y16 -- nullable -- the same values times `scale` as ONE fp16 tensor)
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, u16* __restrict__ y, u16* __restrict__ y16, float scale, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const float4 a = *reinterpret_cast<const float4*>(x + i * 8), b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        store8(y, i * 8, n8 * 8, f);
        if constexpr (X3) { if (y16 != nullptr) *reinterpret_cast<uint4*>(y16 + i * 8) = pack8_half(f, scale); }
    }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const u16* __restrict__ x, float* __restrict__ y, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float f[8];
        load8(x, i * 8, n8 * 8, f);
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

// (zt != nullptr: the blocks past `pblocks` zero the step's accumulators -- the rows {address, 32-bit words} of sar_zero_ranges' table, every
// block striding over every range -- so that the two launches every train step begins with are one: each costs ~4.5 us before it does
// anything)
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackRow* __restrict__ tab, int n, int pblocks, const long long* __restrict__ zt, int nz) {
    if ((int)blockIdx.x >= pblocks) {
        const long long zb = blockIdx.x - pblocks, nzb = gridDim.x - pblocks;
        for (int r = 0; r < nz; ++r) {
            uint4* p = reinterpret_cast<uint4*>(zt[2 * r]);
            const long long nw = zt[2 * r + 1], n4 = nw >> 2;
            for (long long i = zb * 256 + threadIdx.x; i < n4; i += 256LL * nzb) p[i] = make_uint4(0u, 0u, 0u, 0u);
            if (zb == 0 && threadIdx.x < (nw & 3)) reinterpret_cast<unsigned*>(p)[n4 * 4 + threadIdx.x] = 0u;
        }
        return;
    }
    // the row of this block: rows are sorted by first_block; every lane tests one row (one load round trip for up to 64 rows -- a scan
    // was up to n dependent L2 round trips in front of the last convolutions' blocks, most of this kernel's 25 us)
    int r = -1;
    for (int base = 0; base < n; base += 64) {
        const int q = base + (threadIdx.x & 63);
        const bool le = q < n && tab[q].first_block <= (long)blockIdx.x;
        r += __popcll(__ballot(le));
    }
    r = __builtin_amdgcn_readfirstlane(r);
    const PackRow row = tab[r];
    const int taps = (int)row.taps, Cin = (int)row.Cin, Cout = (int)row.Cout;
    // (SAR_X3: the stream interleaves the hi and the lo fragment of every 16-channel block.  row.pad_ & 1, SAR_X3 only -- the mixed backward of
    // round 5: the INPUT-GRADIENT stream is a single plane of IEEE fp16 fragments, the stream sar_conv2d_f16a reads)
    const bool d16 = X3 && (row.pad_ & 1);
    const int npd = d16 ? 1 : NPL;
    const int Ff = NPL * taps * Cin / 16, Ffp = Ff + 8, Fd = npd * taps * Cout / 16, Fdp = Fd + 8;
    const long nf = (long)(Cout / 32) * Ffp * 64, nd = (long)(Cin / 32) * Fdp * 64;    // 16-byte pieces
    long i = ((long)blockIdx.x - row.first_block) * 256 + threadIdx.x;
    if (i >= nf + nd) return;
    const bool fwd = i < nf;
    if (!fwd) i -= nf;
    const int Fp = fwd ? Ffp : Fdp, F = fwd ? Ff : Fd, Cred = fwd ? Cin : Cout;
    const int npl = (fwd || !d16) ? NPL : 1;
    const int lane = (int)(i & 63);
    const long t = i >> 6;
    const int f = (int)(t % Fp), nt = (int)(t / Fp);
    // (SAR_X3: the thread of the hi fragment writes the lo fragment -- the next one, 64 pieces on -- as well; the threads of the odd fragments have
    // nothing to do: gathering the same eight values twice was half of this launch's loads)
    if (npl == 2 && (f & 1)) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (f < F) {
        const int fl = f / npl, nk = Cred / 16, tap = fl / nk, kc = fl % nk;
        const int oc = nt * 32 + (lane & 31), rc = kc * 16 + 8 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = fwd ? row.w[((long)tap * Cin + rc + j) * Cout + oc] : row.w[((long)(taps - 1 - tap) * Cin + oc) * Cout + rc + j];
    }
    u16* dst = (fwd ? row.wf : row.wd) + i * 8;
    if (X3 && npl == 1) {                                            // the fp16 input-gradient stream of the mixed backward
        *reinterpret_cast<uint4*>(dst) = pack8_half(v, 1.f);
        return;
    }
    uint4 o = pack8(v);
    if constexpr (X3) {
        uint4 lo;
        split8(v, o, lo);
        *reinterpret_cast<uint4*>(dst + 64 * 8) = lo;                // fragment f + 1 of the same channel block (F and the padding are even)
    }
    *reinterpret_cast<uint4*>(dst) = o;
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

static int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v;
    }
    return n;
}

template <typename KT>
static bool allow_lds(KT kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}
}  // namespace

extern "C" {

// mirrors BnP (include/sarnet.h: sarBnRef)
typedef BnP sarBnRef;

static int g_last_class[6];
static int g_last_planned = 0;
// sar_conv2d_pair_* runs sar_conv2d_* twice with this set: the launcher then hands back what it would have launched instead of launching it
struct ConvLaunch { ConvP p; dim3 grid; int nthr; size_t lds; int cls[6]; int act32; };
static ConvLaunch* g_capture = nullptr;

// Plan tables live in a workspace the CALLER lends for the life of the process (this library never allocates): sar_conv_plan_workspace_*.
// A shape's tables are written by conv_plan_kernel, on the stream of the launch that first meets the shape and in front of it; later launches
// find them (host-side list, one caller thread like g_last_class).  Without a workspace -- or once it is full -- blocks derive everything
// themselves, as before.
struct PlanKey { int v[16]; };
struct PlanEntry { PlanKey key; size_t thr, tile, lane; int pl_ts; };
static unsigned char* g_plan_ws = nullptr;
static size_t g_plan_bytes = 0, g_plan_used = 0;
static PlanEntry g_plans[256];
static int g_nplans = 0;
static void conv_plan_attach(ConvP& p, int TM, int nthr, int ntiles, hipStream_t stream) {
    p.pl_thr = nullptr; p.pl_tile = nullptr; p.pl_lane = nullptr; p.pl_ts = 0; p.rpi = p.cw > 0 ? nthr / p.cw : 0;
    p.ent_bins = p.fb.bins; p.ent_bbins = p.fb.bbins; p.ent_det = p.fb.det; p.cap_rows_e = p.cap_rows; p.fill_e = p.fill; p.tmv_e = p.tmv;
    static const bool off = getenv("SAR_CONV_PLAN") != nullptr && atoi(getenv("SAR_CONV_PLAN")) == 0;
    if (g_plan_ws == nullptr || off || (p.cap_rows + 1) * 8 > ROWTAB_BYTES || p.cap_rows + 1 > nthr) return;
    const Geo& g = p.g;
    PlanKey k = {{g.B, g.Hs, g.Ws, g.Cin, g.Ho, g.Wo, g.K, g.s, g.d, g.pt, g.pl, p.tmv, TM, nthr, p.cap_rows, p.rs}};
    const PlanEntry* e = nullptr;
    for (int i = 0; i < g_nplans && e == nullptr; ++i)
        if (memcmp(&g_plans[i].key, &k, sizeof(k)) == 0) e = &g_plans[i];
    if (e == nullptr) {
        if (g_nplans >= 256) return;
        const int pl_ts = 2 + 2 * p.cap_rows;
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t b_thr = up((size_t)nthr * 16), b_tile = up((size_t)ntiles * pl_ts * 4), b_lane = up((size_t)ntiles * TM * 4);
        if (g_plan_used + b_thr + b_tile + b_lane > g_plan_bytes) return;
        PlanEntry& n = g_plans[g_nplans];
        n.key = k; n.thr = g_plan_used; n.tile = n.thr + b_thr; n.lane = n.tile + b_tile; n.pl_ts = pl_ts;
        PlanArgs a;
        memset(&a, 0, sizeof(a));
        a.g = g; a.lg_pc = p.lg_pc; a.rs = p.rs; a.cw = p.cw; a.mg_cw = p.mg_cw; a.nthr = nthr; a.tmv = p.tmv; a.TM = TM; a.cap_rows = p.cap_rows;
        a.ntiles = ntiles; a.pl_ts = pl_ts;
        a.thr = reinterpret_cast<uint4*>(g_plan_ws + n.thr); a.tile = reinterpret_cast<unsigned*>(g_plan_ws + n.tile);
        a.lane = reinterpret_cast<unsigned*>(g_plan_ws + n.lane);
        hipLaunchKernelGGL(conv_plan_kernel, dim3(ntiles + 1), dim3(256), 0, stream, a);
        if (hipGetLastError() != hipSuccess) return;
        g_plan_used += b_thr + b_tile + b_lane; ++g_nplans;
        e = &n;
    }
    p.pl_thr = reinterpret_cast<const uint4*>(g_plan_ws + e->thr); p.pl_tile = reinterpret_cast<const unsigned*>(g_plan_ws + e->tile);
    p.pl_lane = reinterpret_cast<const unsigned*>(g_plan_ws + e->lane); p.pl_ts = e->pl_ts;
}
// buf: device memory, 256-byte aligned, that stays valid (and is not written by anyone else) until the next call of this function or the end of
// the process; bytes = 0 / buf = NULL detaches.  A few MB hold every shape of a model (~100 KB per conv shape at B = 64).
int SAR_T16(sar_conv_plan_workspace_)(void* buf, long bytes) {
    if (bytes < 0 || (buf == nullptr && bytes != 0) || ((size_t)buf & 255)) return SAR_EINVAL;
    g_plan_ws = bytes > 0 ? static_cast<unsigned char*>(buf) : nullptr;
    g_plan_bytes = bytes > 0 ? (size_t)bytes : 0; g_plan_used = 0; g_nplans = 0;
    return SAR_OK;
}
// <WN, NKC, VT, KG, WMX, BR> of the most recent sar_conv2d_* launch of this process (introspection for bench.py / tools; not thread-safe)
int SAR_T16(sar_conv2d_last_class_)(int* out6) {
    if (out6 == nullptr) return SAR_EINVAL;
    for (int i = 0; i < 6; ++i) out6[i] = g_last_class[i];
    out6[5] |= g_last_planned << 1;                                   // bit 1 (round 6): the planned instantiation (template argument PL) ran
    return SAR_OK;
}

int SAR_T16(sar_conv_pack_weights_)(const void* table, int n, int total_blocks, hipStream_t stream) {
    if (table == nullptr || n < 1 || total_blocks < 1) return SAR_EINVAL;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(total_blocks), dim3(256), 0, stream, static_cast<const PackRow*>(table), n, total_blocks,
                       static_cast<const long long*>(nullptr), 0);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
// sar_zero_ranges(zero_table, nz) and sar_conv_pack_weights_*(table, n, total_blocks) in ONE launch: what a train step of the 2-byte trunk begins with.
int SAR_T16(sar_conv_pack_weights_zero_)(const void* table, int n, int total_blocks, const void* zero_table, int nz, hipStream_t stream) {
    if (table == nullptr || n < 1 || total_blocks < 1 || zero_table == nullptr || nz < 1 || nz > 65535) return SAR_EINVAL;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(total_blocks + 2048), dim3(256), 0, stream, static_cast<const PackRow*>(table), n, total_blocks,
                       static_cast<const long long*>(zero_table), nz);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// Generic 2-byte convolution: y[M][Cout] = sum over taps, channels of T(src)[padded position] * wpk (+ bias) (+ res), with
//   fill 0: T = identity; 1: T = relu(bn(.)) of `fbn` (forward statistics from fbn->bins, or fbn->saved when bins is null);
//        2: T = BatchNorm(+ReLU) backward of (da = src, y = src2) through `fbn`;  3: the same + src3 (the shortcut gradient).
//   side (nullable): T(src) written back to HBM (shape of src) -- the plain operand of the matching weight gradient.
//   epi  0: none; 1: sum / sum of squares of y into e_bins; 2: BatchNorm-backward sums of y (= da) against (e_y, e_saved); 3: epi 1 in fixed point.
// Forward: d = 1, s = stride, pads as TF.  Input gradient: src = dy, d = stride, s = 1, pt = K-1-pad_top, wpk = dgrad stream.
// SAR_X3 (sar_conv2d_x3): every 2-byte tensor is a plane pair, `side` ONE fp16 tensor = fp16(T(src) * side_scale), side2 (nullable) T(src) as planes.
// act32 (the fp16 build only; sar_conv2d_f16a): src2 and e_y -- tensors of the FORWARD pass -- are fp32, everything else this build's 2-byte type.
#if SAR_HALF
#define A32V true
#else
#define A32V false                                                    // (no such instantiations in the other builds: the table rows alias the plain ones)
#endif
static int conv2d_impl(const void* src, const void* src2, const void* src3, const void* wpk, const float* bias, const void* res, void* y, void* side,
                       int B, int Hs, int Ws, int Cin, int Ho, int Wo, int Cout, int K, int s, int d, int pt, int pl, int fill,
                       const sarBnRef* fbn, int epi, float* e_bins, const void* e_y, const float* e_saved, void* side2, float side_scale, hipStream_t stream,
                       int act32 = 0) {
    ConvP p;
    memset(&p, 0, sizeof(p));
    if (act32 && !SAR_HALF) return SAR_EINVAL;
    const int a32 = act32 ? 1 : 0;
    if (!make_geo(p.g, B, Hs, Ws, Cin, Ho, Wo, K, s, d, pt, pl)) return SAR_EINVAL;
    if (!pow2(Cin) || !pow2(Cout) || Cin < 32 || Cout < 32 || Cin > CMAX || Cout > CMAX) return SAR_EINVAL;
    if (fill < 0 || fill > 3 || epi < 0 || epi > 3 || (fill && fbn == nullptr) || (fill >= 2 && src2 == nullptr) || (fill == 3 && src3 == nullptr)) return SAR_EINVAL;
    if (epi && e_bins == nullptr) return SAR_EINVAL;
    if (epi == 2 && (e_y == nullptr || e_saved == nullptr)) return SAR_EINVAL;
    if ((long)B * Hs * Ws * Cin >= (1L << 31) / NPL || (long)p.g.M * Cout >= (1L << 31) / NPL) return SAR_EINVAL;
    if (!X3 && side2 != nullptr) return SAR_EINVAL;
    p.Cout = Cout;
    p.x = static_cast<const u16*>(src); p.x2 = static_cast<const u16*>(src2); p.x3 = static_cast<const u16*>(src3); p.wpk = static_cast<const h16x8*>(wpk);
    p.bias = bias; p.res = static_cast<const u16*>(res); p.y = static_cast<u16*>(y); p.side = static_cast<u16*>(side);
    p.side2 = static_cast<u16*>(side2); p.side_scale = side_scale != 0.f ? side_scale : 1.f;
    p.fill = fill; if (fbn) p.fb = *fbn;
    p.epi = epi; p.e_bins = e_bins; p.e_y = static_cast<const u16*>(e_y); p.e_saved = e_saved;
    p.lg_pc = ilog2(Cin / 8); p.rs = Cin * 2 + 16;
    p.cw = p.g.Wp * (Cin / 8); p.mg_cw = magic(p.cw);
    p.Fs = NPL * K * K * Cin / 16 + 8;
    // (SAR_X3, 32 source channels into 64 at stride 1 -- the input gradients of the first block's conv and shortcut, 128 000 pixels at B = 64:
    // as TWO 32-channel tiles of the co-resident eight-wave form, 1 000 blocks in two rounds, instead of 2 000 four-wave blocks of 64 x 64 in
    // four: 71 + 25 -> 37 + 20 us)
    const bool split64 = X3 && Cin == 32 && Cout == 64 && s == 1 && d == 1 && getenv("SAR_CONV16_TM") == nullptr;
    const int WN = (Cout >= 64 && !split64) ? 2 : 1, TN = 32 * WN;
    // LDS bytes of the patch: every plane + a 1 KiB scratch slot behind it (the fill's branch-free no-op pieces land there)
    auto plane_bytes = [&](int tm) { return (size_t)patch_rows_bound(p.g, tm < p.g.M ? tm : p.g.M) * p.g.Wp * p.rs + 1024; };
    // Pixel tile of the 32-output-channel launches (stage 0, its shortcut and gradients).  Round 2 ran them as 128-pixel tiles on four
    // waves: 1 000 blocks at B = 64 for 768 resident slots -- 1.3 rounds, the second a third full.  256 pixels on EIGHT waves (128
    // registers, two blocks per CU) are 500 blocks = ONE round: the ten stage-0 launches 171 -> 128 us.  Taken where two such blocks fit
    // a CU's LDS.  (64 pixels on two waves went the other way, 169 -> 270 us: twice the blocks pay the fixed prologue / statistics phases,
    // 1.5x the halo rows.)  SAR_X3: the doubled patch admits ONE eight-wave block per CU, or two four-wave blocks of 128 pixels; the step
    // measures 2.586 ms with the former, 2.637 with the latter: taken wherever it fits at all.  SAR_CONV16_TM = 128 | 256 forces one form.
    static const int tm_env = getenv("SAR_CONV16_TM") ? atoi(getenv("SAR_CONV16_TM")) : 0;
    const bool narrow = WN == 1 && Cin <= 64;
    bool tm256 = false;
    if (narrow && (tm_env == 0 || tm_env == 256)) {
        const size_t pb = NPL * plane_bytes(256);
        tm256 = tm_env == 256 || (size_t)6 * Cin * 4 + ROWTAB_BYTES * 2 + ((pb + 1023) & ~(size_t)1023) + 3 * 8 * 1024 <= (X3 ? 160 : 80) * 1024;
    }
    // BR form (weight fragments in registers, no ring): 64+ source channels into 64+ output channels -- stages 1-3
    const bool br = WN == 2 && Cin >= 64;
    // sources whose 64-pixel patch does not fit the LDS (res_filters = 64: the 512-channel stage exceeds 160 KB by 1.3 KB, its stride-2 entry
    // from 256 channels by more; SAR_X3, twice the patch: the 256-channel stage as well): 32-pixel tiles
    // (Negative, round 4 -- VERDICT r3 item 2a: half tiles for the 128-channel sources, 32 pixels on FOUR waves (2 channel halves x 2 k-groups,
    // four fragments per tap and wave) so that a stage-2 launch is 500 blocks, two co-resident per CU whose serial phases could overlap:
    // 2.6525 vs 2.6507 ms per step in split precision, 1.956 vs 1.936 in fp16.  Twice the blocks pay the fixed prologue; not kept.)
    bool br32 = false;
    if (br) {
        const size_t pb64 = NPL * plane_bytes(64);
        br32 = Cin == 512 || (size_t)6 * Cin * 4 + ROWTAB_BYTES + ((pb64 + 1023) & ~(size_t)1023) > 160 * 1024;   // (512: the 32-pixel form is the only instantiation)
    }
    const int TM = br32 ? 32 : tm256 ? 256 : 32 * (4 / WN);
    // split precision, sources whose 32-pixel patch still does not fit (res_filters = 64: the 512-channel stage and its stride-2 entry from 256
    // channels): the 32-pixel form on SIXTEEN owned pixels per block
    p.tmv = TM;
    if (X3 && br32 && (size_t)6 * Cin * 4 + ROWTAB_BYTES + ((NPL * plane_bytes(32) + 1023) & ~(size_t)1023) > 160 * 1024) p.tmv = 16;
    // SAR_X3, the four-wave ring forms of the 32-channel sources (stage 1's stride-2 entry and its 1x1 shortcut: the patch of 64 output pixels is
    // 17 + rows of the twice-as-large input, 2 x 30 KB): as many owned pixels -- whole output rows -- as leave room for TWO blocks per CU
    // (with one, four waves have a CU to themselves and the pair launch is four rounds)
    // (the same for the four-wave BR form of the 64-channel sources where its patch is past half a CU: the stride-2 entry of stage 2, 106 KB)
    if (X3 && ((!br && !tm256 && Cin == 32) || (br && !br32 && Cin == 64)) && p.g.Wo > 0 && p.g.Wo <= TM) {
        auto lds_of = [&](int tmv) { return (size_t)6 * Cin * 4 + ROWTAB_BYTES + ((NPL * plane_bytes(tmv) + 1023) & ~(size_t)1023) + (br ? 0 : (size_t)3 * WN * 4 * 1024); };
        if (lds_of(TM) > 80 * 1024) {
            int t = TM / p.g.Wo * p.g.Wo;
            while (t > TM / 2 && lds_of(t) > 80 * 1024) t -= p.g.Wo;
            if (t > TM / 2 && lds_of(t) <= 80 * 1024) p.tmv = t;
        }
    }
    // SAR_X3, BR forms, a launch that does not fill the chip (stage 3 at B = 64: 39 pixel tiles x 4 channel tiles = 156 blocks on 256 CUs; every
    // stage at smaller batches): FEWER owned pixels per block, as many pixel tiles as one round holds.  The MFMA rows past the owned pixels
    // recompute the neighbour's and are dropped (the k-phase does not shrink), but the fill, the epilogue and the statistics -- two thirds of a
    // block's life -- do, and the idle CUs take the extra tiles: stage 3 at 39 owned pixels = 256 blocks, 22.3 -> 21.1 us forward, 25-27 -> 23.4-24
    // backward (a block's fixed latency chain, not its pixel count, is most of its life: DESIGN 5.00).
    // (round 5: every build -- the mixed backward's input gradients of stage 3 run in the fp16 build)
    if (br && p.tmv == TM) {
        const long tiles = cdiv(p.g.M, TM), nty = Cout / TN, cus = cu_count();
        if (tiles * nty * 20 <= cus * 17) {                                        // at most 85 % of one round
            const long slots = cus / nty;
            const int t = slots > 0 ? (int)cdiv(p.g.M, (int)slots) : TM;
            if (t < TM && t >= 8) p.tmv = t;
        }
    }
    p.cap_rows = patch_rows_bound(p.g, p.tmv < p.g.M ? p.tmv : p.g.M);
    if (p.cap_rows > 256 || p.cap_rows * 8 > ROWTAB_BYTES) return SAR_EINVAL;
    p.lplane = (int)plane_bytes(p.tmv);
    size_t patch_bytes = (size_t)NPL * p.lplane;
    const int KG = br ? (Cin >= 128 ? 4 : 2) : 1;                                  // k-groups (BR: eight waves at 128+ source channels, four at 64)
    const int nthr = br ? 128 * KG : tm256 ? 512 : 256;
    if (p.cap_rows > nthr) return SAR_EINVAL;                                      // one thread per row of the row table
    const size_t stage_bytes = (size_t)KG * TM * TN * 4 + (size_t)nthr / (TN / SAR_EPI_EV) * TN * 2 * 8;   // partial rows: 8 bytes per entry (epi 3)
    // the long k-loops of the 128+-channel ring layers get a 4-deep weight ring where it fits
    p.rd = (Cin >= 128 && (size_t)6 * Cin * 4 + ROWTAB_BYTES + ((patch_bytes + 1023) & ~(size_t)1023) + (size_t)4 * WN * 8 * 1024 <= 160 * 1024) ? 4 : 3;
    const int SF = (X3 && !br && Cin == 32) ? 4 : 8;                               // fragments per ring stage (conv_body: CO2)
    const size_t ring_bytes = br ? 0 : (size_t)p.rd * WN * SF * 1024;
    if (patch_bytes + ring_bytes < stage_bytes) patch_bytes = stage_bytes - ring_bytes;   // the staging tile overlays patch + ring
    if (patch_bytes < (size_t)nthr * 32) patch_bytes = (size_t)nthr * 32;      // bn_finish's scratch: G x 4 x Cin x 8 bytes, G x Cin <= threads
    if (Cin > nthr || p.cw > nthr) return SAR_EINVAL;   // (a patch row wider than the block: no such layer in the model's range)
    patch_bytes = (patch_bytes + 1023) & ~(size_t)1023;
    p.patch_bytes = (int)patch_bytes;
    const size_t lds = (size_t)6 * Cin * 4 + ROWTAB_BYTES + patch_bytes + ring_bytes;
    if (lds > 160 * 1024) return SAR_EINVAL;
    const dim3 grid((unsigned)cdiv(p.g.M, p.tmv), (unsigned)(Cout / TN));
    conv_plan_attach(p, TM, nthr, (int)grid.x, stream);
    const int pln = p.pl_thr != nullptr ? 1 : 0;                                  // the planned instantiation (ConvP: plan tables)
    g_last_planned = pln;
    // template arguments of the instantiation this launch runs, as rocprofv3 prints them (bench.py classes its conv timings by them)
    g_last_class[0] = WN; g_last_class[1] = Cin / 16 < 8 ? Cin / 16 : 8; g_last_class[2] = K * K * (Cin / 128 > 1 ? Cin / 128 : 1);
    g_last_class[3] = KG; g_last_class[4] = br32 ? 1 : br ? 2 : tm256 ? 8 : 4 / WN; g_last_class[5] = br ? 1 : 0;
    if (g_capture != nullptr) {
        g_capture->p = p; g_capture->grid = grid; g_capture->nthr = nthr; g_capture->lds = lds; g_capture->act32 = a32;
        for (int i = 0; i < 6; ++i) g_capture->cls[i] = g_last_class[i];
        return SAR_OK;
    }
    // instantiations: [WN][NKC in {2,4,8}][K == 3]; Cin = 256 / 512 run NKC = 8 with 2 / 4 channel parts per tap
    typedef void (*kern_t)(const ConvP);
    const int ki = ilog2(Cin / 16) - 1, k3 = K == 3 ? 1 : 0;
    if (br) {
        // 64-pixel tiles: 64 | 128 | 256 source channels; 32-pixel tiles: 64 | 128 | 256 | 512
        // [planned][act32][class][K == 3]
#define CVB(A_, P_, NKC_, NH_, KG_, WMX_) {conv_bf16_kernel<2, NKC_, NH_, KG_, WMX_, true, A_, P_>, conv_bf16_kernel<2, NKC_, 9 * NH_, KG_, WMX_, true, A_, P_>}
#define CVB7(A_, P_) {CVB(A_, P_, 4, 1, 2, 2), CVB(A_, P_, 8, 1, 4, 2), CVB(A_, P_, 8, 2, 4, 2), CVB(A_, P_, 4, 1, 2, 1), CVB(A_, P_, 8, 1, 4, 1), CVB(A_, P_, 8, 2, 4, 1), CVB(A_, P_, 8, 4, 4, 1)}
        static const kern_t kernsb[2][2][7][2] = {{CVB7(false, false), CVB7(A32V, false)}, {CVB7(false, true), CVB7(A32V, true)}};
#undef CVB
#undef CVB7
        static PerDeviceFlag attrb[2][2][7][2];
        const int kb = br32 ? ki + 2 : ki - 1;
        const kern_t kern = kernsb[pln][a32][kb][k3];
        if (!attrb[pln][a32][kb][k3]) { if (!allow_lds(kern, 160 * 1024)) return SAR_ELAUNCH; attrb[pln][a32][kb][k3] = true; }
        hipLaunchKernelGGL(kern, grid, dim3(nthr), lds, stream, p);
        SAR_CHECK_LAUNCH();
        return SAR_OK;
    }
    if (tm256) {
#define CVX(A_, P_, NKC_, WMX_) {conv_bf16_kernel<1, NKC_, 1, 1, WMX_, false, A_, P_>, conv_bf16_kernel<1, NKC_, 9, 1, WMX_, false, A_, P_>}
#define CVX2(A_, P_) {CVX(A_, P_, 2, 8), CVX(A_, P_, 4, 8)}
        static const kern_t kernsx[2][2][2][2] = {{CVX2(false, false), CVX2(A32V, false)}, {CVX2(false, true), CVX2(A32V, true)}};
#undef CVX
#undef CVX2
        static PerDeviceFlag attrx[2][2][2][2];
        const kern_t kern = kernsx[pln][a32][ki][k3];
        if (!attrx[pln][a32][ki][k3]) { if (!allow_lds(kern, 160 * 1024)) return SAR_ELAUNCH; attrx[pln][a32][ki][k3] = true; }
        hipLaunchKernelGGL(kern, grid, dim3(nthr), lds, stream, p);
        SAR_CHECK_LAUNCH();
        return SAR_OK;
    }
#define CV(A_, P_, WN_, NKC_, NH_) {conv_bf16_kernel<WN_, NKC_, NH_, 1, 4 / WN_, false, A_, P_>, conv_bf16_kernel<WN_, NKC_, 9 * NH_, 1, 4 / WN_, false, A_, P_>}
#define CV10(A_, P_) {{CV(A_, P_, 1, 2, 1), CV(A_, P_, 1, 4, 1), CV(A_, P_, 1, 8, 1), CV(A_, P_, 1, 8, 2), CV(A_, P_, 1, 8, 4)}, \
                      {CV(A_, P_, 2, 2, 1), CV(A_, P_, 2, 4, 1), CV(A_, P_, 2, 8, 1), CV(A_, P_, 2, 8, 2), CV(A_, P_, 2, 8, 4)}}
    static const kern_t kerns[2][2][2][5][2] = {{CV10(false, false), CV10(A32V, false)}, {CV10(false, true), CV10(A32V, true)}};
#undef CV
#undef CV10
    static PerDeviceFlag attr[2][2][2][5][2];
    const kern_t kern = kerns[pln][a32][WN - 1][ki][k3];
    if (!attr[pln][a32][WN - 1][ki][k3]) { if (!allow_lds(kern, 160 * 1024)) return SAR_ELAUNCH; attr[pln][a32][WN - 1][ki][k3] = true; }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

#if SAR_X3
int sar_conv2d_x3(const void* src, const void* src2, const void* src3, const void* wpk, const float* bias, const void* res, void* y, void* side,
                  int B, int Hs, int Ws, int Cin, int Ho, int Wo, int Cout, int K, int s, int d, int pt, int pl, int fill,
                  const sarBnRef* fbn, int epi, float* e_bins, const void* e_y, const float* e_saved, void* side2, float side_scale, hipStream_t stream) {
    return conv2d_impl(src, src2, src3, wpk, bias, res, y, side, B, Hs, Ws, Cin, Ho, Wo, Cout, K, s, d, pt, pl, fill, fbn, epi, e_bins, e_y, e_saved, side2, side_scale, stream);
}
#else
int SAR_T16(sar_conv2d_)(const void* src, const void* src2, const void* src3, const void* wpk, const float* bias, const void* res, void* y, void* side,
                    int B, int Hs, int Ws, int Cin, int Ho, int Wo, int Cout, int K, int s, int d, int pt, int pl, int fill,
                    const sarBnRef* fbn, int epi, float* e_bins, const void* e_y, const float* e_saved, hipStream_t stream) {
    return conv2d_impl(src, src2, src3, wpk, bias, res, y, side, B, Hs, Ws, Cin, Ho, Wo, Cout, K, s, d, pt, pl, fill, fbn, epi, e_bins, e_y, e_saved, nullptr, 1.f, stream);
}
#endif

// The arguments of one sar_conv2d_* call (include/sarnet.h: sarConvArgs; side2 / side_scale: sar_conv2d_x3 only, 0 elsewhere).
struct sarConvArgs { const void *src, *src2, *src3, *wpk; const float* bias; const void* res; void* y; void* side;
                     int B, Hs, Ws, Cin, Ho, Wo, Cout, K, s, d, pt, pl, fill; const sarBnRef* fbn; int epi; float* e_bins; const void* e_y; const float* e_saved;
                     void* side2; float side_scale; };
// Two independent convolutions in ONE launch: a = a 3x3 convolution, b = a 1x1 convolution of the same channel classes and block size (the first
// conv of a shape-changing residual block and its shortcut, or their input gradients).  Returns SAR_EINVAL -- and launches NOTHING -- when the
// two do not run on instantiations that pair up; the caller then issues them one after the other.
static int conv2d_pair_impl(const sarConvArgs* a, const sarConvArgs* b, int act32, hipStream_t stream);
int SAR_T16(sar_conv2d_pair_)(const sarConvArgs* a, const sarConvArgs* b, hipStream_t stream) { return conv2d_pair_impl(a, b, 0, stream); }
#if SAR_HALF
// The mixed backward of the split-precision mode (round 5): sar_conv2d_f16 / sar_conv2d_pair_f16 with src2 and e_y -- the tensors of the forward pass
// a gradient launch reads (model.py:197 trains in fp32; resnet.py:22-26 the BatchNorm whose backward they feed) -- as the fp32 tensors sar_conv2d_x3 wrote.
int sar_conv2d_f16a(const void* src, const void* src2, const void* src3, const void* wpk, const float* bias, const void* res, void* y, void* side,
                    int B, int Hs, int Ws, int Cin, int Ho, int Wo, int Cout, int K, int s, int d, int pt, int pl, int fill,
                    const sarBnRef* fbn, int epi, float* e_bins, const void* e_y, const float* e_saved, hipStream_t stream) {
    return conv2d_impl(src, src2, src3, wpk, bias, res, y, side, B, Hs, Ws, Cin, Ho, Wo, Cout, K, s, d, pt, pl, fill, fbn, epi, e_bins, e_y, e_saved, nullptr, 1.f, stream, 1);
}
int sar_conv2d_pair_f16a(const sarConvArgs* a, const sarConvArgs* b, hipStream_t stream) { return conv2d_pair_impl(a, b, 1, stream); }
#endif
static int conv2d_pair_impl(const sarConvArgs* a, const sarConvArgs* b, int act32, hipStream_t stream) {
    if (a == nullptr || b == nullptr) return SAR_EINVAL;
    static ConvLaunch la, lb;                                                       // (host-side scratch; one caller thread, like g_last_class)
    const sarConvArgs* q[2] = {a, b};
    ConvLaunch* l[2] = {&la, &lb};
    for (int i = 0; i < 2; ++i) {
        g_capture = l[i];
        const int rc = conv2d_impl(q[i]->src, q[i]->src2, q[i]->src3, q[i]->wpk, q[i]->bias, q[i]->res, q[i]->y, q[i]->side, q[i]->B, q[i]->Hs, q[i]->Ws,
                                   q[i]->Cin, q[i]->Ho, q[i]->Wo, q[i]->Cout, q[i]->K, q[i]->s, q[i]->d, q[i]->pt, q[i]->pl, q[i]->fill, q[i]->fbn, q[i]->epi,
                                   q[i]->e_bins, q[i]->e_y, q[i]->e_saved, q[i]->side2, q[i]->side_scale, stream, act32);
        g_capture = nullptr;
        if (rc != SAR_OK) return rc;
    }
    const int* ca = la.cls; const int* cb = lb.cls;
    // same WN, NKC, KG, WMX, BR and block size; a: nine taps, b: one
    // (or, BR forms: a on 32-pixel tiles, b on 64-pixel ones -- `mixed`)
    const bool mixed = ca[5] && ca[4] == 1 && cb[4] == 2;
    if (ca[0] != cb[0] || ca[1] != cb[1] || ca[3] != cb[3] || (ca[4] != cb[4] && !mixed) || ca[5] != cb[5] || la.nthr != lb.nthr || ca[2] != 9 * cb[2]) return SAR_EINVAL;
    typedef void (*pair_t)(const ConvP, const ConvP, unsigned, unsigned, unsigned, unsigned);
    pair_t kern = nullptr;
    static PerDeviceFlag at[28];
    int ki = -1;
    const bool a32 = act32 != 0, pl = la.p.pl_thr != nullptr && lb.p.pl_thr != nullptr;   // (both planned, or neither: one kernel runs both bodies)
    if (!pl) { la.p.pl_thr = nullptr; lb.p.pl_thr = nullptr; }
#define PK(...) (pl ? (a32 ? conv_bf16_pair_kernel<__VA_ARGS__, A32V, true> : conv_bf16_pair_kernel<__VA_ARGS__, false, true>) \
                    : (a32 ? conv_bf16_pair_kernel<__VA_ARGS__, A32V, false> : conv_bf16_pair_kernel<__VA_ARGS__, false, false>))
#define PKB(...) (pl ? (a32 ? conv_bf16_pair_kernel<__VA_ARGS__, A32V, true, 2> : conv_bf16_pair_kernel<__VA_ARGS__, false, true, 2>) \
                     : (a32 ? conv_bf16_pair_kernel<__VA_ARGS__, A32V, false, 2> : conv_bf16_pair_kernel<__VA_ARGS__, false, false, 2>))
    if (mixed) {
        // the forward entry of the last stage at res_filters 32 (128 -> 256 channels, stride 2): 21.6 + 7.3 us as two launches
        if (ca[0] == 2 && ca[1] == 8 && ca[2] == 9 && ca[3] == 4) { kern = PKB(2, 8, 9, 4, 1, true); ki = 24 + (a32 ? 1 : 0) + (pl ? 2 : 0); }
    } else
    if (ca[0] == 2 && ca[1] == 2 && ca[2] == 9 && ca[3] == 1 && ca[4] == 2 && !ca[5]) { kern = PK(2, 2, 9, 1, 2, false); ki = 0; }
    else if (ca[0] == 2 && ca[1] == 4 && ca[2] == 9 && ca[3] == 2 && ca[4] == 2 && ca[5]) { kern = PK(2, 4, 9, 2, 2, true); ki = 1; }
    else if (ca[0] == 2 && ca[1] == 8 && ca[2] == 9 && ca[3] == 4 && ca[4] == 2 && ca[5]) { kern = PK(2, 8, 9, 4, 2, true); ki = 2; }
    else if (ca[0] == 2 && ca[1] == 8 && ca[2] == 18 && ca[3] == 4 && ca[4] == 2 && ca[5]) { kern = PK(2, 8, 18, 4, 2, true); ki = 3; }
    else if (ca[0] == 2 && ca[1] == 8 && ca[2] == 18 && ca[3] == 4 && ca[4] == 1 && ca[5]) { kern = PK(2, 8, 18, 4, 1, true); ki = 4; }
    // (round 5) the 256-pixel form of the 64-channel sources: res34's stem is 64 wide whatever res_filters is (resnet.py:191), so at res_filters 32
    // the first block's conv and its 1x1 shortcut both read the pooled map -- 30 + 14.6 us as two launches
    else if (ca[0] == 1 && ca[1] == 4 && ca[2] == 9 && ca[3] == 1 && ca[4] == 8 && !ca[5]) { kern = PK(1, 4, 9, 1, 8, false); ki = 5; }
#undef PK
#undef PKB
    if (a32 && ki >= 0 && ki < 24) ki += 6;
    if (pl && ki >= 0 && ki < 24) ki += 12;                                         // (24-27: the mixed form carries its own)
    if (kern == nullptr) return SAR_EINVAL;
    if (!at[ki]) { if (!allow_lds(kern, 160 * 1024)) return SAR_ELAUNCH; at[ki] = true; }
    const unsigned na = la.grid.x * la.grid.y, nb = lb.grid.x * lb.grid.y, na8 = (na + 7u) & ~7u;
    const size_t lds = la.lds > lb.lds ? la.lds : lb.lds;
    hipLaunchKernelGGL(kern, dim3(na8 + nb), dim3(la.nthr), lds, stream, la.p, lb.p, na, la.grid.y, nb, lb.grid.y);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

#if !SAR_X3
// One convolution of a batched weight-gradient launch as the caller describes it (include/sarnet.h: sarWgradConv).
struct sarWgradConv { const void* x; const void* dy; float* dw; float* dbias; int B, H, W, Cin, Ho, Wo, Cout, K, s, pt, pl; float ginv; };

#if !SAR_HALF && !SAR_X3
// Host-only planning (no device access; shared by both builds of this file): the device tables of ONE launch over `n` convolutions of equal K (1 or 3) and equal
// channel-tile class -- convs_out: n rows of 128 bytes, items_out: up to max_items rows of 32 bytes, longest first -- plus the
// launch parameters launch4 = {dynamic LDS bytes, positions of the patch ring (3x3), chunk buffers, tile_ci << 4 | tile_co}.
// tile_ci / tile_co (1 or 2): 32-channel planes per block on the input / output side (3x3 only; every Cin must be a multiple
// of 32*tile_ci, every Cout of 32*tile_co).  Runs of chunks are cut so that the launch has about `target_blocks` items.
int sar_wgrad_plan_bf16(const sarWgradConv* convs, int n, int target_blocks, int tile_ci, int tile_co, void* convs_out, void* items_out,
                        int max_items, int* n_items, int* launch4, void* tiles_out, int* slices_out) {
    if (convs == nullptr || n < 1 || n > 1024 || target_blocks < 1 || convs_out == nullptr || items_out == nullptr || n_items == nullptr || launch4 == nullptr ||
        tiles_out == nullptr || slices_out == nullptr)
        return SAR_EINVAL;
    const int K = convs[0].K;
    if (K != 1 && K != 3) return SAR_EINVAL;
    if (tile_ci < 1 || tile_ci > 2 || tile_co < 1 || tile_co > 2 || (K == 1 && (tile_ci != 1 || tile_co != 1))) return SAR_EINVAL;
    // pixels per chunk (wgrad3_body: CH), per CONVOLUTION: 128 since round 6 where the convolution's positions fit the 512-position ring with a
    // chunk in flight, else 64 as in rounds 2-5 (SAR_WG_CH=64 forces that, for A/B runs).  Bit 8 of launch4[3]: some convolution of the plan runs
    // 128-pixel chunks (the launcher then takes the kernel that holds both bodies).
    static const int ch_env = getenv("SAR_WG_CH") ? atoi(getenv("SAR_WG_CH")) : 128;
    const int NWV = (tile_ci == 2 && tile_co == 2) ? 12 : 9;
    WgConv tmp;
    unsigned char* co = static_cast<unsigned char*>(convs_out);
    long total = 0;                                                  // 64-pixel chunk x channel-tile units of the whole launch
    auto prep = [&](int i, int ch) -> bool {                         // validates descriptor i and builds its table row in tmp
        const sarWgradConv& c = convs[i];
        memset(&tmp, 0, sizeof(tmp));
        if (c.K != K || !make_geo(tmp.g, c.B, c.H, c.W, c.Cin, c.Ho, c.Wo, c.K, c.s, 1, c.pt, c.pl)) return false;
        if (!pow2(c.Cin) || !pow2(c.Cout) || c.Cin < 32 * tile_ci || c.Cout < 32 * tile_co || c.Cin > CMAX || c.Cout > CMAX) return false;
        if (c.x == nullptr || c.dy == nullptr || c.dw == nullptr) return false;
        if ((long)c.B * c.H * c.W * c.Cin >= (1L << 31) || (long)tmp.g.M * c.Cout >= (1L << 31)) return false;
        tmp.Cout = c.Cout;
        tmp.x = static_cast<const u16*>(c.x); tmp.dy = static_cast<const u16*>(c.dy); tmp.dw = c.dw; tmp.dbias = c.dbias;
        tmp.ch = (short)ch;
        tmp.nchunks = cdiv(tmp.g.M, ch);
        tmp.cap_pos = K == 1 ? 64 : (patch_rows_bound(tmp.g, ch < tmp.g.M ? ch : tmp.g.M) * tmp.g.Wp + 15) & ~15;
        tmp.la = 3;
        tmp.ginv = c.ginv != 0.f ? c.ginv : 1.f;                        // (0: descriptors written before the field existed)
        // 3x3: DMA instructions of an item's first chunk (the whole patch) dealt to the block's waves: at most 4 (128-pixel chunks: 6) each
        return K == 1 || (ch / 16) * tile_co + (tmp.cap_pos / 16 + 1) * tile_ci <= (ch == 64 ? 4 : 6) * NWV;
    };
    // 3x3: the ring must hold the padded positions of la + 1 consecutive chunks (the one being multiplied and la in flight) plus
    // the 16-position alignment at both ends.  64-pixel chunks: 512 positions when every convolution then gets la >= 2, otherwise the smallest
    // power of two with la >= 1; every convolution takes the deepest la <= 3 that fits.  128-pixel chunks: where 512 positions hold two of them
    // (la = 1; la = 2 -- what three 64-pixel chunks in flight were -- where they hold three).
    auto need = [](const Geo& g, int la, int ch) { const int px = (la + 1) * ch; return patch_rows_bound(g, px < g.M ? px : g.M) * g.Wp + 32; };
    int np = 512, la_max = 1, any128 = 0, front_max = 0;
    static unsigned char chs[1024];
    for (int i = 0; i < n; ++i) {
        int ch = 64;
        if (K == 3 && ch_env != 64 && prep(i, 128) && need(tmp.g, 1, 128) <= 512) ch = 128;
        chs[i] = (unsigned char)ch;
        if (!prep(i, ch)) return SAR_EINVAL;
        if (K == 3 && ch == 64)
            while (need(tmp.g, np <= 512 ? 2 : 1, 64) > np) { np *= 2; if (np > 2048) return SAR_EINVAL; }
    }
    for (int i = 0; i < n; ++i) {
        const int ch = chs[i];
        prep(i, ch);
        if (K == 3) {
            tmp.la = ch == 128 ? 2 : 3;
            while (tmp.la > 1 && need(tmp.g, tmp.la, ch) > np) --tmp.la;
            if (need(tmp.g, tmp.la, ch) > np) return SAR_EINVAL;
            if (tmp.la > la_max) la_max = tmp.la;
            if (ch == 128) any128 = 1;
            const int front = 8 * ch + (tmp.la + 1) * tile_co * ch * 64;    // position tables + gradient-row slots of this convolution's blocks
            if (front > front_max) front_max = front;
        }
        memset(co + (size_t)i * 128, 0, 128);
        memcpy(co + (size_t)i * 128, &tmp, sizeof(tmp));
        total += (long)cdiv(tmp.g.M, 64) * (convs[i].Cin / (32 * tile_ci)) * (convs[i].Cout / (32 * tile_co));
    }
    int maxc = (int)((total + target_blocks - 1) / target_blocks);   // 64-pixel chunks per item
    static const int minc3 = getenv("SAR_WG_MINC") ? atoi(getenv("SAR_WG_MINC")) : 16;   // chunks per item at least: an item ends in its tiles' atomics (8 / 16 / 24 / 32 / 48: 2.327 / 2.307 / 2.317 / 2.331 / 2.365 ms, forced one-rank three-graph step)
    const int minc = K == 3 ? minc3 : 2 * WG_NW1;
    if (maxc < minc) maxc = minc;
    WgItem* items = static_cast<WgItem*>(items_out);
    int ni = 0, tab_pos = 0;
    for (int i = 0; i < n; ++i) {
        const sarWgradConv& c = convs[i];
        const int CH = chs[i], maxc_i = maxc * 64 / CH > 0 ? maxc * 64 / CH : 1;
        const int nchunks = cdiv((long)c.B * c.Ho * c.Wo, CH);
        int slices = cdiv(nchunks, maxc_i), cpb = cdiv(nchunks, slices);
        if (K == 3) {
            // the item's position -> source-offset table (4 bytes per padded position of its run): at most WG_TAB_MAX positions
            prep(i, CH);
            auto run_pos = [&](int chunks) { const long px = (long)chunks * CH; return patch_rows_bound(tmp.g, (int)(px < tmp.g.M ? px : tmp.g.M)) * tmp.g.Wp + 48; };
            while (cpb > 1 && run_pos(cpb) > WG_TAB_MAX) { ++slices; cpb = cdiv(nchunks, slices); }
            if (run_pos(cpb) > WG_TAB_MAX) return SAR_EINVAL;
            if (run_pos(cpb) > tab_pos) tab_pos = run_pos(cpb);
        }
        for (int sl = 0; sl * cpb < nchunks; ++sl)
            for (int ci = 0; ci < c.Cin; ci += 32 * tile_ci)
                for (int cc = 0; cc < c.Cout; cc += 32 * tile_co) {
                    if (ni >= max_items) return SAR_EINVAL;
                    WgItem it;
                    memset(&it, 0, sizeof(it));
                    it.conv = i; it.ci0 = ci; it.co0 = cc; it.c_first = sl * cpb;
                    it.c_count = nchunks - sl * cpb < cpb ? nchunks - sl * cpb : cpb;
                    it.single = cpb >= nchunks ? 1 : 0;
                    items[ni++] = it;
                }
    }
    // longest first (the hardware dispatches blocks in index order); stable, so the tiles of one pixel run stay adjacent
    for (int a = 1; a < ni; ++a) {
        const WgItem it = items[a];
        int b = a - 1;
        while (b >= 0 && items[b].c_count * (int)chs[items[b].conv] < it.c_count * (int)chs[it.conv]) { items[b + 1] = items[b]; --b; }   // (by pixels: chunk sizes differ)
        items[b + 1] = it;
    }
    *n_items = ni;
    // partial-tile slabs and the reduction table: an item that shares its tile gets the next slab; tiles are listed in (convolution,
    // channel tile) order with their slices in pixel order, so the reduction's sum order is fixed by the plan
    {
        WgTile* tiles = static_cast<WgTile*>(tiles_out);
        int nslab = 0, ntile = 0, nsl = 0;
        for (int a = 0; a < ni; ++a) items[a].slab = items[a].single ? -1 : nslab++;
        for (int i = 0; i < n; ++i)
            for (int ci = 0; ci < convs[i].Cin; ci += 32 * tile_ci)
                for (int cc = 0; cc < convs[i].Cout; cc += 32 * tile_co) {
                    WgTile t;
                    memset(&t, 0, sizeof(t));
                    t.conv = i; t.ci0 = ci; t.co0 = cc; t.first = nsl;
                    // (quadratic in the worst case, a few thousand items: microseconds, once per plan)
                    int last_first = -1;
                    for (;;) {                                                   // slices in increasing c_first
                        int best = -1;
                        for (int a = 0; a < ni; ++a)
                            if (!items[a].single && items[a].conv == i && items[a].ci0 == ci && items[a].co0 == cc && items[a].c_first > last_first &&
                                (best < 0 || items[a].c_first < items[best].c_first)) best = a;
                        if (best < 0) break;
                        slices_out[nsl++] = items[best].slab; last_first = items[best].c_first;
                    }
                    t.count = nsl - t.first;
                    if (t.count > 0) tiles[ntile++] = t;
                }
        launch4[4] = ntile; launch4[5] = nslab; launch4[6] = (K == 3 ? 9 : 1) * tile_ci * tile_co * 1024; launch4[7] = 0;
    }
    if (K == 3) {
        tab_pos = (tab_pos + 127) & ~127;
        launch4[0] = front_max + tab_pos * 4 + tile_ci * np * 64; launch4[1] = np; launch4[2] = la_max + 1;
    } else {
        launch4[0] = WG_NW1 * WG_NBUF * 8192; launch4[1] = 0; launch4[2] = WG_NBUF;
    }
    launch4[3] = (tile_ci << 4) | tile_co | (any128 ? 0x100 : 0) | ((K == 3 ? tab_pos / 128 : 0) << 16);
    return launch4[0] <= 160 * 1024 ? SAR_OK : SAR_EINVAL;
}

#endif  // !SAR_HALF && !SAR_X3

// Weight gradients of many convolutions, one launch: tables and launch4 from sar_wgrad_plan_bf16 (tables uploaded by the caller),
// K = kernel size of the planned convolutions.  dw += x^T (*) dy per tap, dbias (where given) += column sums of dy.
int SAR_T16(sar_conv2d_wgrad_batch_)(const void* convs_dev, const void* items_dev, int n_items, int K, const int* launch4, const void* tiles_dev,
                                     const int* slices_dev, float* slab, hipStream_t stream) {
    if (convs_dev == nullptr || items_dev == nullptr || launch4 == nullptr || n_items < 1 || (K != 1 && K != 3)) return SAR_EINVAL;
    const int lds_bytes = launch4[0], np = launch4[1], nbuf = launch4[2], tci = (launch4[3] >> 4) & 15, tco = launch4[3] & 15, mix = (launch4[3] >> 8) & 1;
    const int tab_pos = (launch4[3] >> 16) * 128;
    const int n_tiles = launch4[4], n_slabs = launch4[5];
    if (n_tiles < 0 || n_slabs < 0 || (n_slabs > 0 && (n_tiles < 1 || tiles_dev == nullptr || slices_dev == nullptr || slab == nullptr))) return SAR_EINVAL;
    if (lds_bytes <= 0 || lds_bytes > 160 * 1024 || tci < 1 || tci > 2 || tco < 1 || tco > 2) return SAR_EINVAL;
    static const uint4* zero_page = nullptr;
    if (zero_page == nullptr) {
        void* zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_page)) != hipSuccess) return SAR_ELAUNCH;
        zero_page = static_cast<const uint4*>(zp);
    }
    if (K == 3) {
        const int tab_bytes = tab_pos * 4;
        // (the front of a block's LDS -- position tables and gradient-row slots -- is per convolution: WgConv::ch, la; the plan's launch4[0] holds the largest)
        if (!pow2(np) || np < 512 || nbuf < 2 || nbuf > 4 || tab_bytes > WG_TAB_MAX * 4 || lds_bytes < 512 + 2 * tco * 4096 + tab_bytes + tci * np * 64) return SAR_EINVAL;
        if (launch4[6] != 9 * tci * tco * 1024) return SAR_EINVAL;
        Wg3Args a{static_cast<const unsigned char*>(convs_dev), static_cast<const WgItem*>(items_dev), zero_page, (unsigned)np, nbuf, tab_bytes / 4, slab};
        typedef void (*k3_t)(const Wg3Args);
        // 64 x 64 tiles: the twelve-wave form
        static const k3_t k3[2][2][2] = {{{wgrad3_bf16_kernel<1, 1>, wgrad3_bf16_kernel<1, 2>}, {wgrad3_bf16_kernel<2, 1>, wgrad3_bf16_kernel<2, 2, true>}},
                                         {{wgrad3_bf16_kernel<1, 1, false, true>, wgrad3_bf16_kernel<1, 2, false, true>},
                                          {wgrad3_bf16_kernel<2, 1, false, true>, wgrad3_bf16_kernel<2, 2, true, true>}}};
        static PerDeviceFlag attr3[2][2][2];
        const k3_t kern = k3[mix][tci - 1][tco - 1];
        PerDeviceFlag& at = attr3[mix][tci - 1][tco - 1];
        if (!at) { if (!allow_lds(kern, 160 * 1024)) return SAR_ELAUNCH; at = true; }
        const int threads = tci == 2 && tco == 2 ? 768 : 576;
        hipLaunchKernelGGL(kern, dim3(n_items), dim3(threads), lds_bytes, stream, a);
        if (n_tiles > 0 && !launch4[7])                                             // launch4[7]: the caller reduces (sar_wgrad_reduce_multi_*)
            hipLaunchKernelGGL(wgrad_reduce_kernel<true>, dim3(n_tiles, 9 * tci * tco), dim3(256), 0, stream, static_cast<const unsigned char*>(convs_dev),
                               static_cast<const WgTile*>(tiles_dev), slices_dev, slab, tci, tco);
    } else {
        if (tci != 1 || tco != 1 || mix || nbuf != WG_NBUF || lds_bytes != WG_NW1 * WG_NBUF * 8192) return SAR_EINVAL;
        static PerDeviceFlag attr1;
        if (!attr1) { if (!allow_lds(wgrad1_bf16_kernel, 160 * 1024)) return SAR_ELAUNCH; attr1 = true; }
        hipLaunchKernelGGL(wgrad1_bf16_kernel, dim3(n_items), dim3(WG_NW1 * 64), lds_bytes, stream, static_cast<const unsigned char*>(convs_dev),
                           static_cast<const WgItem*>(items_dev), zero_page, slab);
        if (n_tiles > 0 && !launch4[7])
            hipLaunchKernelGGL(wgrad_reduce_kernel<false>, dim3(n_tiles, 1), dim3(256), 0, stream, static_cast<const unsigned char*>(convs_dev),
                               static_cast<const WgTile*>(tiles_dev), slices_dev, slab, 1, 1);
    }
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// One reduction launch for up to four batched weight-gradient launches that ran with launch4[7] = 1 (mirrors include/sarnet.h: sarWgradReduce).
struct sarWgradReduce { const void* convs_dev; const void* tiles_dev; const int* slices_dev; const float* slab; int n_tiles, K, tile_ci, tile_co; };
int SAR_T16(sar_wgrad_reduce_multi_)(const sarWgradReduce* plans, int n, hipStream_t stream) {
    if (plans == nullptr || n < 1 || n > 4) return SAR_EINVAL;
    RedPlans P;
    memset(&P, 0, sizeof(P));
    int blocks = 0;
    for (int i = 0; i < 4; ++i) {
        P.first[i] = blocks;
        if (i >= n) continue;
        const sarWgradReduce& r = plans[i];
        const int K = r.K & 255, wr = (r.K >> 8) & 1;                                 // K + 256: the sums are written (dw = ...), not added
        if (r.n_tiles < 0 || (K != 1 && K != 3) || (r.K & ~0x1ff) || r.tile_ci < 1 || r.tile_ci > 2 || r.tile_co < 1 || r.tile_co > 2) return SAR_EINVAL;
        if (r.n_tiles > 0 && (!r.convs_dev || !r.tiles_dev || !r.slices_dev || !r.slab)) return SAR_EINVAL;
        P.p[i] = RedPlan{static_cast<const unsigned char*>(r.convs_dev), static_cast<const WgTile*>(r.tiles_dev), r.slices_dev, r.slab, r.n_tiles,
                         (K == 3 ? 1 : 0) | (wr << 1), r.tile_ci, r.tile_co};
        blocks += r.n_tiles * (K == 3 ? 9 * r.tile_ci * r.tile_co : 1);
    }
    P.first[4] = blocks;
    for (int i = n; i < 4; ++i) P.first[i] = blocks;                                // (no block falls into an absent plan)
    if (blocks == 0) return SAR_OK;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), 0, stream, P);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

#endif  // !SAR_X3

static int bn_grid(long R, int C) {
    const long pieces = R * (C / 8);
    long nb = (pieces + 256 * 4 - 1) / (256 * 4);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// out_scale: multiplies the result (the fp16 trunk's static gradient scale enters here, at the final BatchNorm; 1 otherwise).
// sar_bn_bwd_apply_x3: `out` (a plane pair) stays in true units; out16 (nullable) = fp16(result * out_scale), the weight gradients' operand.
static int bn_bwd_apply_impl(const void* dact, int dact_f32, const void* x, const sarBnRef* bn, const void* add, void* out, void* out16, long R,
                             int C, int relu, float out_scale, hipStream_t stream) {
    if (bn == nullptr || !pow2(C) || C < 8 || C > CMAX || R < 1) return SAR_EINVAL;
    const dim3 grid(bn_grid(R, C));
    if (dact_f32)
        hipLaunchKernelGGL(bn_bwd_apply_bf16_kernel<float>, grid, dim3(256), 0, stream, static_cast<const float*>(dact), static_cast<const u16*>(x), *bn,
                           static_cast<const u16*>(add), static_cast<u16*>(out), static_cast<u16*>(out16), R, C, relu, out_scale);
    else
        hipLaunchKernelGGL(bn_bwd_apply_bf16_kernel<u16>, grid, dim3(256), 0, stream, static_cast<const u16*>(dact), static_cast<const u16*>(x), *bn,
                           static_cast<const u16*>(add), static_cast<u16*>(out), static_cast<u16*>(out16), R, C, relu, out_scale);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
#if SAR_X3
int sar_bn_bwd_apply_x3(const void* dact, int dact_f32, const void* x, const sarBnRef* bn, const void* add, void* out, void* out16, long R,
                        int C, int relu, float out_scale, hipStream_t stream) {
    return bn_bwd_apply_impl(dact, dact_f32, x, bn, add, out, out16, R, C, relu, out_scale, stream);
}
#else
int SAR_T16(sar_bn_bwd_apply_)(const void* dact, int dact_f32, const void* x, const sarBnRef* bn, const void* add, void* out, long R,
                          int C, int relu, float out_scale, hipStream_t stream) {
    return bn_bwd_apply_impl(dact, dact_f32, x, bn, add, out, nullptr, R, C, relu, out_scale, stream);
}
#endif

int SAR_T16(sar_bn_bwd_reduce_)(const void* dact, int dact_f32, const void* x, const float* saved, float* bbins, long R, int C, int relu,
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

#if SAR_HALF
#define sar_bn_apply_T16_f32 sar_bn_apply_f16_f32
#define sar_cast_T16_f32 sar_cast_f16_f32
#elif SAR_X3
#define sar_bn_apply_T16_f32 sar_bn_apply_x3_f32
#define sar_cast_T16_f32 sar_cast_x3_f32
#else
#define sar_bn_apply_T16_f32 sar_bn_apply_bf16_f32
#define sar_cast_T16_f32 sar_cast_bf16_f32
#endif
int sar_bn_apply_T16_f32(const void* x, const sarBnRef* bn, float* out, long R, int C, int relu, hipStream_t stream) {
    if (bn == nullptr || !pow2(C) || C < 8 || C > CMAX || R < 1) return SAR_EINVAL;
    hipLaunchKernelGGL(bn_apply_bf16_f32_kernel, dim3(bn_grid(R, C)), dim3(256), 0, stream, static_cast<const u16*>(x), *bn, out, R, C, relu);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// fp32 -> the trunk's 2-byte type (sar_cast_f32_x3: a plane pair of n elements each, + y16 (nullable) = fp16(x * scale))
#if SAR_X3
int sar_cast_f32_x3(const float* x, void* y, void* y16, float scale, long n, hipStream_t stream) {
#else
int SAR_T16(sar_cast_f32_)(const float* x, void* y, long n, hipStream_t stream) {
    void* y16 = nullptr; const float scale = 1.f;
#endif
    if (n < 8 || n % 8) return SAR_EINVAL;
    long nb = (n / 8 + 1023) / 1024; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, static_cast<u16*>(y), static_cast<u16*>(y16), scale, n / 8);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
int sar_cast_T16_f32(const void* x, float* y, long n, hipStream_t stream) {
    if (n < 8 || n % 8) return SAR_EINVAL;
    long nb = (n / 8 + 1023) / 1024; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)nb), dim3(256), 0, stream, static_cast<const u16*>(x), y, n / 8);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

#if !SAR_HALF && !SAR_X3
int sar_bn_bins(void) { return NB; }
#endif

#ifdef SAR_TIMING
int sar_dbg_set_timing_buffer16(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tbuf16), &buf, sizeof(buf)) == hipSuccess ? SAR_OK : SAR_ELAUNCH;
}
#endif

}  // extern "C"
