// The accent classifier of SAR-Net in ONE launch, forward + loss + backward:
//   AR_CF_DS1 = Dense(64, relu) -> AR_CF_DS2 = Dense(64, relu) -> y_accent = Dense(A, softmax)   (model.py:150-152 via DS, model.py:35-42)
//   loss = Keras categorical_crossentropy(y_true, y_accent)                                      (model.py:350)
// On a 64-utterance batch these are 64 x 256 x 64, 64 x 64 x 64 and 64 x 64 x 8 products: as separate launches (3 GEMMs, softmax,
// cross-entropy forward; cross-entropy / softmax backward, 6 GEMMs, 2 activation backwards) they were 15 launches of ~5-14 us each
// around ~1 us of arithmetic -- 110 us of a 2.4 ms step spent on dispatch latency.  Here a workgroup owns four rows and walks the whole
// chain with the intermediates in LDS; the weight gradients of the rows leave as fp32 atomics.
// fp32 FMA arithmetic throughout (also in the "bf16" mode, whose GEMMs would round these operands).
#include "heads_margin.h"

namespace {

constexpr int FH_ROWS = 4;      // rows per workgroup: 16 workgroups for the reference's batch of 64 -- ONE workgroup of 64 rows ran 180 us
                                // (3.5 M multiply-adds at the 64 per cycle of a single CU, operands from L2), slower than the launches it replaced
constexpr int FH_DMAX = 64;     // max width of the two hidden layers
constexpr int FH_AMAX = 32;     // max classes
constexpr int FH_NT = 1024;    // sixteen waves: the loops are chains of L2 loads, and waves are what hides them on a CU that runs one block
constexpr float FH_KEPS = 1e-7f;

struct FusedHeadP {
    const float* x; const float* W1; const float* b1; const float* W2; const float* b2; const float* W3; const float* b3;
    const float* y_true; const float* dloss;
    float* h1; float* h2; float* probs; float* loss;
    float* dW1; float* db1; float* dW2; float* db2; float* dW3; float* db3; float* dx;
    int B, H, D1, D2, A;
    // MG (round 5): the margin-softmax head on the same input rows (losses.py:28-47 / 74-91 / 121-144; heads_margin.h), same A classes
    const float* my; const float* myt; const float* mdl; const float* Wm;      // one-hot input, y_true, loss seed, kernel (H,A)
    float* mprobs; float* mloss; float* minvw; float* dWm; float* mws;         // mws: H*A + 1 words, zero on entry and on exit
    int mmode; float ms, mm;
};
constexpr int FH_MG_H = 256, FH_MG_A = 8;      // the margin head's register-cached form (heads_margin.h: CACHE)

// MG: the workgroup also runs the margin head's training step (forward, cross-entropy, backward: heads_margin.h) of its four rows, as
// slices of work in the classifier's own phases -- no barrier is added:
//   before barrier 1   wave (row r = w / 4, slice q = w % 4) holds x[r][64 q + lane] and kernel row 64 q + lane in registers and reduces its
//                      slice's share of |x|^2, |W[:,a]|^2, x . W[:,a]  (the one-wave-per-sample kernel was ~2 400 dependent instructions)
//   barrier 1 -> 2     waves 4..7: the class-level arithmetic of one row each (margin_row_math), beside the four waves of the first product
//   barrier 2 -> 3     every wave: its slice of d loss / d x_normalised and of the projection; all threads: the rows' sum of the
//                      normalised-kernel gradient, H*A atomics per workgroup into mws (in flight until barrier 6)
//   barrier 3 -> 4     the rows' input gradients into LDS (they meet the classifier's in its last loop: dx = both, one plain store)
//   barrier 6 -> end   ticket (drawn under the y_accent phase, looked at under AR_CF_DS2's); the LAST workgroup to draw one takes the batch's
//                      sum back out of mws (atomic exchange: leaves zeros) under AR_CF_DS1's phase and applies the column-normalisation
//                      backward (margin_wbwd_body) behind it.
// Three launches (28.5 + 20 + 5 us in the step) become one.
template <bool MG>
__global__ __launch_bounds__(FH_NT) void accent_classifier_kernel(const FusedHeadP p) {
    constexpr int MA = FH_MG_A, MQ = FH_MG_H / 64;
    __shared__ float mpart[MG ? FH_ROWS : 1][MG ? MQ : 1][2 * MA + 1], mdc[MG ? FH_ROWS : 1][MA], miw[MG ? FH_ROWS : 1][MA], mix[FH_ROWS], mproj[FH_ROWS][MQ];
    __shared__ float mdx[MG ? FH_ROWS : 1][MG ? FH_MG_H : 1], mred[2][MA];
    __shared__ int lastf;
    __shared__ float h1s[FH_ROWS][FH_DMAX + 1], h2s[FH_ROWS][FH_DMAX + 1], g1s[FH_ROWS][FH_DMAX + 1], g2s[FH_ROWS][FH_DMAX + 1];
    __shared__ float z3s[FH_ROWS][FH_AMAX + 1], g3s[FH_ROWS][FH_AMAX + 1];
    __shared__ float pq[4][FH_ROWS][FH_DMAX + 1];                   // the four k-quarters of the first product
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    const int H = p.H, D1 = p.D1, D2 = p.D2, A = p.A, P1 = D1 + 1, P2 = D2 + 1, P3 = A + 1;
    float* W1s = dyn;                  // [H][D1 + 1]
    float* W2s = W1s + H * P1;         // [D1][D2 + 1]
    float* W3s = W2s + D1 * P2;        // [D2][A + 1]
    float* xs = W3s + D2 * P3;         // [FH_ROWS][H]: the block's input rows
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * FH_ROWS, nr = min(FH_ROWS, p.B - r0);
    const bool train = p.y_true != nullptr && p.dloss != nullptr;
    // margin head: this thread's element d of row mr, and its kernel row
    const int lane = tid & 63, mr = (tid >> 6) / MQ, mq = (tid >> 6) % MQ, md = mq * 64 + lane;
    const bool mrow = MG && mr < nr, mon = mrow && md < H;
    float mxv = 0.f, mwv[MA], mdxn = 0.f;
    if constexpr (MG) {
        if (tid < 2 * MA) mred[tid / MA][tid % MA] = 0.f;             // (read after barrier 7)
        float4 w0 = make_float4(0, 0, 0, 0), w1 = w0;
        if (mon) {
            mxv = p.x[(long)(r0 + mr) * H + md];
            w0 = *reinterpret_cast<const float4*>(p.Wm + (long)md * A);
            if (A > 4) w1 = *reinterpret_cast<const float4*>(p.Wm + (long)md * A + 4);
        }
        const float ws[MA] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int a = 0; a < MA; ++a) mwv[a] = ws[a];
    }
    // (waves 4..7 own the class-level arithmetic of one row each after barrier 1: its inputs are requested here)
    const int m2r = (tid >> 6) - FH_ROWS;
    const bool m2 = MG && m2r >= 0 && m2r < nr;
    float myy = 0.f, myt = 0.f, mgl = 0.f;
    if (m2) {
        if (lane < A) { myy = p.my[(long)(r0 + m2r) * A + lane]; myt = p.myt[(long)(r0 + m2r) * A + lane]; }
        mgl = p.mdl[r0 + m2r];
    }

    // Everything the chain reads from global memory, once, with all loads in flight: the three kernels (82 KB) and the block's rows.
    // (Read where they are used -- in the phase loops, from L2 -- every phase began with a memory round trip nothing could hide on a CU
    // that runs one block: 129 us with one load pair per trip, 51 us unrolled 16 deep.)
    for (int i = tid; i < H * D1; i += FH_NT) W1s[(i / D1) * P1 + i % D1] = p.W1[i];
    for (int i = tid; i < D1 * D2; i += FH_NT) W2s[(i / D2) * P2 + i % D2] = p.W2[i];
    for (int i = tid; i < D2 * A; i += FH_NT) W3s[(i / A) * P3 + i % A] = p.W3[i];
    for (int i = tid; i < FH_ROWS * H; i += FH_NT) xs[i] = p.x[(long)(r0 + min(i / H, nr - 1)) * H + i % H];
    if constexpr (MG) {
        if (mrow) {                                                  // wave-uniform
            const float sx = wave_sum(mxv * mxv);
            if (lane == 0) mpart[mr][mq][2 * MA] = sx;
#pragma unroll
            for (int a = 0; a < MA; ++a) {
                const float sw = wave_sum(mwv[a] * mwv[a]), dot = wave_sum(mxv * mwv[a]);
                if (lane == 0) { mpart[mr][mq][a] = sw; mpart[mr][mq][MA + a] = dot; }
            }
        }
    }
    __syncthreads();
    if constexpr (MG) {
        if (m2) {
            const int r = m2r, a = min(lane, MA - 1), b = r0 + r;
            float sx = 0.f, sw = 0.f, dot = 0.f;
            for (int q = 0; q < MQ; ++q) { sx += mpart[r][q][2 * MA]; sw += mpart[r][q][a]; dot += mpart[r][q][MA + a]; }
            const float ix = 1.f / sqrtf(fmaxf(sx, 1e-12f)), iw = 1.f / sqrtf(fmaxf(sw, 1e-12f));
            const float c = lane < A ? dot * ix * iw : 0.f;
            const float dc = margin_row_math(b, lane, A, p.mmode, p.ms, p.mm, c, myy, myt, mgl, p.mprobs, p.mloss);
            if (lane < A) {
                mdc[r][lane] = dc; miw[r][lane] = iw;
                if (b == 0) p.minvw[lane] = sw >= 1e-12f ? iw : -iw;
            } else if (lane < MA) { mdc[r][lane] = 0.f; miw[r][lane] = 0.f; }
            if (lane == 0) mix[r] = sx < 1e-12f ? -ix : ix;             // (sign: the projection is dropped for a zero row, as in margin_train_body)
        }
    }
    // ---- h1 = relu(x W1 + b1): thread = (column j, quarter q of the k range), all four rows per thread; the quarters meet in LDS
    for (int o = tid; o < D1 * 4; o += FH_NT) {
        const int j = o % D1, q = o / D1;
        const int k0 = (H * q) / 4, k1 = (H * (q + 1)) / 4;
        float a[FH_ROWS];
#pragma unroll
        for (int r = 0; r < FH_ROWS; ++r) a[r] = 0.f;
#pragma unroll 8
        for (int k = k0; k < k1; ++k) {
            const float wv = W1s[k * P1 + j];
#pragma unroll
            for (int r = 0; r < FH_ROWS; ++r) a[r] = fmaf(xs[r * H + k], wv, a[r]);
        }
#pragma unroll
        for (int r = 0; r < FH_ROWS; ++r) pq[q][r][j] = a[r];
    }
    __syncthreads();
    if constexpr (MG) {
        if (mrow) {
            const float ix = fabsf(mix[mr]);
#pragma unroll
            for (int a = 0; a < MA; ++a) mdxn = fmaf(mdc[mr][a], mwv[a] * miw[mr][a], mdxn);
            const float pj = wave_sum(mxv * ix * mdxn);
            if (lane == 0) mproj[mr][mq] = pj;
        }
        // d loss / d (normalised kernel), summed over the block's rows; the batch's sum collects in mws
        for (int i = tid; i < H * A; i += FH_NT) {
            const int d = i / A, a = i % A;
            float v = 0.f;
            for (int r = 0; r < nr; ++r) v = fmaf(xs[r * H + d] * fabsf(mix[r]), mdc[r][a], v);
            atomicAdd(p.mws + i, v);
        }
    }
    for (int o = tid; o < D1 * FH_ROWS; o += FH_NT) {
        const int j = o % D1, r = o / D1;
        h1s[r][j] = fmaxf((pq[0][r][j] + pq[1][r][j]) + (pq[2][r][j] + pq[3][r][j]) + p.b1[j], 0.f);
    }
    __syncthreads();
    if constexpr (MG) {
        if (mon) {
            const float ixs = mix[mr], ix = fabsf(ixs);
            float proj = 0.f;
            for (int q = 0; q < MQ; ++q) proj += mproj[mr][q];
            if (ixs < 0.f) proj = 0.f;
            mdx[mr][md] = (mdxn - mxv * ix * proj) * ix;
        }
    }
    // ---- h2 = relu(h1 W2 + b2)
    for (int o = tid; o < D2 * FH_ROWS; o += FH_NT) {
        const int j = o % D2, r = o / D2;
        float a = p.b2[j];
#pragma unroll 8
        for (int k = 0; k < D1; ++k) a = fmaf(h1s[r][k], W2s[k * P2 + j], a);
        h2s[r][j] = fmaxf(a, 0.f);
    }
    __syncthreads();
    // ---- logits, softmax, cross-entropy (+ its gradient w.r.t. the logits)
    for (int o = tid; o < A * FH_ROWS; o += FH_NT) {
        const int a_ = o % A, r = o / A;
        float a = p.b3[a_];
#pragma unroll 8
        for (int k = 0; k < D2; ++k) a = fmaf(h2s[r][k], W3s[k * P3 + a_], a);
        z3s[r][a_] = a;
    }
    for (int o = tid; o < FH_ROWS * D1; o += FH_NT) if (o / D1 < nr) p.h1[(long)(r0 + o / D1) * D1 + o % D1] = h1s[o / D1][o % D1];
    for (int o = tid; o < FH_ROWS * D2; o += FH_NT) if (o / D2 < nr) p.h2[(long)(r0 + o / D2) * D2 + o % D2] = h2s[o / D2][o % D2];
    __syncthreads();
    if (tid < nr) {
        const int r = tid;
        float* z = z3s[r];
        float m = -INFINITY;
        for (int a = 0; a < A; ++a) m = fmaxf(m, z[a]);
        float s = 0.f;
        for (int a = 0; a < A; ++a) { const float e = expf(z[a] - m); z[a] = e; s += e; }
        const float inv = 1.f / s;
        float S = 0.f;
        for (int a = 0; a < A; ++a) { z[a] *= inv; p.probs[(long)(r0 + r) * A + a] = z[a]; S += z[a]; }
        if (p.y_true != nullptr) {
            const float* y = p.y_true + (long)(r0 + r) * A;
            // Keras categorical_crossentropy on probabilities: renormalise, clip to [1e-7, 1 - 1e-7] (heads.hip: cce_fwd / cce_bwd)
            float l = 0.f, acc = 0.f;
            for (int a = 0; a < A; ++a) {
                const float q = z[a] / S;
                l -= y[a] * logf(fminf(fmaxf(q, FH_KEPS), 1.f - FH_KEPS));
                const float dq = (q >= FH_KEPS && q <= 1.f - FH_KEPS) ? -y[a] / q : 0.f;
                acc = fmaf(dq, q, acc);
            }
            p.loss[r0 + r] = l;
            if (train) {
                const float g = p.dloss[r0 + r];
                float sd = 0.f;
                for (int a = 0; a < A; ++a) {
                    const float q = z[a] / S;
                    const float dq = (q >= FH_KEPS && q <= 1.f - FH_KEPS) ? -y[a] / q : 0.f;
                    const float dpa = g * (dq - acc) / S;
                    g3s[r][a] = dpa;
                    sd = fmaf(dpa, z[a], sd);
                }
                for (int a = 0; a < A; ++a) g3s[r][a] = z[a] * (g3s[r][a] - sd);       // softmax backward
            }
        }
    } else if (tid < FH_ROWS && train) {
        for (int a = 0; a < A; ++a) g3s[tid][a] = 0.f;                              // rows past the batch
    }
    if (!train) return;
    // this thread's mws atomics (issued three phases ago) are performed before the workgroup draws its ticket.  Only device-scope atomics touch
    // mws and the ticket, so no cache write-back is needed (a full fence here costs 1.3 us: it writes the h1 / h2 / probs lines back first)
    if constexpr (MG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* const ticket = reinterpret_cast<int*>(p.mws + H * A);
    int drawn = 0;
    if constexpr (MG) {
        if (tid == FH_NT - 64) drawn = atomicAdd(ticket, 1);         // (a wave with nothing to do in this phase; the value is looked at after barrier 7)
    }
    // ---- y_accent: dW3 += h2^T dz3, db3 += colsum dz3; dz2 = (dz3 W3^T) * [h2 > 0]
    for (int o = tid; o < D2 * A; o += FH_NT) {
        const int a_ = o % A, k = o / A;
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s = fmaf(h2s[r][k], g3s[r][a_], s);
        atomicAdd(p.dW3 + k * A + a_, s);
    }
    if (tid < A) {
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s += g3s[r][tid];
        atomicAdd(p.db3 + tid, s);
    }
    for (int o = tid; o < FH_ROWS * D2; o += FH_NT) {
        const int k = o % D2, r = o / D2;
        float s = 0.f;
        for (int a = 0; a < A; ++a) s = fmaf(g3s[r][a], W3s[k * P3 + a], s);
        g2s[r][k] = h2s[r][k] > 0.f ? s : 0.f;
    }
    __syncthreads();
    if constexpr (MG) {
        if (tid == FH_NT - 64) {
            lastf = drawn == (int)gridDim.x - 1;
            if (lastf) atomicExch(ticket, 0);
        }
    }
    // ---- AR_CF_DS2: dW2 += h1^T dz2, db2; dz1 = (dz2 W2^T) * [h1 > 0]
    for (int o = tid; o < D1 * D2; o += FH_NT) {
        const int j = o % D2, k = o / D2;
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s = fmaf(h1s[r][k], g2s[r][j], s);
        atomicAdd(p.dW2 + k * D2 + j, s);
    }
    if (tid < D2) {
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s += g2s[r][tid];
        atomicAdd(p.db2 + tid, s);
    }
    for (int o = tid; o < FH_ROWS * D1; o += FH_NT) {
        const int k = o % D1, r = o / D1;
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < D2; ++j) s = fmaf(g2s[r][j], W2s[k * P2 + j], s);
        g1s[r][k] = (r < nr && h1s[r][k] > 0.f) ? s : 0.f;
    }
    __syncthreads();
    // the last workgroup to draw its ticket owns the column-normalisation backward (heads_margin.h: margin_wbwd_body) of the whole batch:
    // thread = elements tid + e FH_NT of the (H,A) kernel, all of class a = tid % A (FH_NT % A == 0)
    constexpr int NE = FH_MG_H * FH_MG_A / FH_NT;
    float lwv[NE], lgv[NE];
    const bool last = MG && lastf;
    if (last) {
        float sw = 0.f, pj = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = tid + e * FH_NT;
            lwv[e] = i < H * A ? p.Wm[i] : 0.f;
            lgv[e] = i < H * A ? atomicExch(p.mws + i, 0.f) : 0.f;    // every workgroup's atomics, and zero for the next step
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) { sw = fmaf(lwv[e], lwv[e], sw); pj = fmaf(lwv[e], lgv[e], pj); }
        for (int m = 32; m >= A; m >>= 1) { sw += __shfl_xor(sw, m, 64); pj += __shfl_xor(pj, m, 64); }
        if (lane < A) { atomicAdd(&mred[0][lane], sw); atomicAdd(&mred[1][lane], pj); }
    }
    // ---- AR_CF_DS1: dW1 += x^T dz1, db1; dx = dz1 W1^T (written, not accumulated: it opens the gradient of the embedding)
    for (int o = tid; o < H * D1; o += FH_NT) {
        const int j = o % D1, k = o / D1;
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s = fmaf(xs[r * H + k], g1s[r][j], s);
        atomicAdd(p.dW1 + k * D1 + j, s);
    }
    if (tid < D1) {
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s += g1s[r][tid];
        atomicAdd(p.db1 + tid, s);
    }
    for (int o = tid; o < nr * H; o += FH_NT) {
        const int k = o % H, r = o / H;
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < D1; ++j) s = fmaf(g1s[r][j], W1s[k * P1 + j], s);         // row k at pitch D1 + 1: conflict-free across the lanes
        p.dx[(long)(r0 + r) * H + k] = MG ? s + mdx[r][k] : s;
    }
    if (last) {
        __syncthreads();                                             // (block-uniform) mred complete
        const int a = tid % A;
        const float swa = mred[0][a], iw = 1.f / sqrtf(fmaxf(swa, 1e-12f));
        const float proj = swa >= 1e-12f ? mred[1][a] * iw : 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = tid + e * FH_NT;
            if (i < H * A) atomicAdd(p.dWm + i, (lgv[e] - lwv[e] * iw * proj) * iw);    // (one addend per element: not a read-modify-write round trip)
        }
    }
}

}  // namespace

// include/sarnet.h.  y_true NULL: forward only (h1, h2, probs); dloss NULL: forward + per-sample loss; both: + backward (parameter
// gradients accumulate, dx is written).  D1, D2 <= 64, A <= 32.
extern "C" int sar_accent_classifier_f32(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                                         const float* b3, const float* y_true, const float* dloss, float* h1, float* h2, float* probs,
                                         float* loss, float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3, float* dx,
                                         int B, int H, int D1, int D2, int A, hipStream_t stream) {
    if (!x || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !h1 || !h2 || !probs || B < 1) return SAR_EINVAL;
    if (H < 1 || D1 < 1 || D1 > FH_DMAX || D2 < 1 || D2 > FH_DMAX || A < 1 || A > FH_AMAX) return SAR_EINVAL;
    if (y_true && !loss) return SAR_EINVAL;
    if (y_true && dloss && (!dW1 || !db1 || !dW2 || !db2 || !dW3 || !db3 || !dx)) return SAR_EINVAL;
    FusedHeadP p{x, W1, b1, W2, b2, W3, b3, y_true, dloss, h1, h2, probs, loss, dW1, db1, dW2, db2, dW3, db3, dx, B, H, D1, D2, A,
                 nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0.f};
    const size_t lds = ((size_t)H * (D1 + 1) + (size_t)D1 * (D2 + 1) + (size_t)D2 * (A + 1) + (size_t)FH_ROWS * H) * sizeof(float);
    if (lds > 140 * 1024) return SAR_EINVAL;               // H <= 480 at D1 = 64
    static PerDeviceFlag attr;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(accent_classifier_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) != hipSuccess)
            return SAR_ELAUNCH;
        attr = true;
    }
    hipLaunchKernelGGL(accent_classifier_kernel<false>, dim3(cdiv(B, FH_ROWS)), dim3(FH_NT), lds, stream, p);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// include/sarnet.h.  The training step of BOTH heads on the embedding x (B,H) in one launch: the accent classifier as above (y_true,
// dloss required) and the margin-softmax head `mode` (0 SphereFace, 1 CosFace, 2 ArcFace; scale s, margin m) with kernel Wm (H,A) on the
// same rows: its one-hot input my, labels myt, loss seed mdl -> mprobs (B,A), mloss (B), minvw (A), dWm += ..., and dx = the SUM of the
// two heads' input gradients (written).  mws: H*A + 1 words, zero before the first call; the launch leaves them zero.
// Sizes of the margin head's register-cached form only: A in {4, 8}, H <= 256, Wm 16-byte aligned (else SAR_EINVAL: the caller runs
// sar_accent_classifier_f32 + sar_margin_head_train_f32).
extern "C" int sar_accent_margin_train_f32(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                                           const float* b3, const float* y_true, const float* dloss, float* h1, float* h2, float* probs,
                                           float* loss, float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3, float* dx,
                                           const float* my, const float* myt, const float* mdl, const float* Wm, float* mprobs, float* mloss,
                                           float* minvw, float* dWm, float* mws, int B, int H, int D1, int D2, int A, int mode, float s,
                                           float m, hipStream_t stream) {
    if (!x || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !h1 || !h2 || !probs || !y_true || !dloss || !loss || B < 1) return SAR_EINVAL;
    if (!dW1 || !db1 || !dW2 || !db2 || !dW3 || !db3 || !dx) return SAR_EINVAL;
    if (!my || !myt || !mdl || !Wm || !mprobs || !mloss || !minvw || !dWm || !mws || mode < 0 || mode > 2) return SAR_EINVAL;
    if (H < 1 || H > FH_MG_H || D1 < 1 || D1 > FH_DMAX || D2 < 1 || D2 > FH_DMAX || (A != 4 && A != 8)) return SAR_EINVAL;
    if (reinterpret_cast<uintptr_t>(Wm) & 15) return SAR_EINVAL;
    FusedHeadP p{x, W1, b1, W2, b2, W3, b3, y_true, dloss, h1, h2, probs, loss, dW1, db1, dW2, db2, dW3, db3, dx, B, H, D1, D2, A,
                 my, myt, mdl, Wm, mprobs, mloss, minvw, dWm, mws, mode, s, m};
    const size_t lds = ((size_t)H * (D1 + 1) + (size_t)D1 * (D2 + 1) + (size_t)D2 * (A + 1) + (size_t)FH_ROWS * H) * sizeof(float);
    static PerDeviceFlag attr;                              // (dynamic: <= 90 KB at H = 256; + 21 KB static stays inside the CU's 160 KB)
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(accent_classifier_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) != hipSuccess)
            return SAR_ELAUNCH;
        attr = true;
    }
    hipLaunchKernelGGL(accent_classifier_kernel<true>, dim3(cdiv(B, FH_ROWS)), dim3(FH_NT), lds, stream, p);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
