// Instruction issue-rate microbenchmark for gfx950 (developer tool): cycles per wave64 instruction, one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int N = 16, IT = 2000;
template <int OP> __global__ __launch_bounds__(256) void k(unsigned long long* out, float seed) {
    float a[N]; f32x2 p[N]; unsigned u[N]; bf16x8 m0, m1; f32x16 acc[2];
    for (int i = 0; i < N; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] * 2}; u[i] = (unsigned)(a[i]); }
    for (int i = 0; i < 8; ++i) { m0[i] = (__bf16)(seed + i); m1[i] = (__bf16)(seed - i); }
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0; acc[1][i] = 0; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if constexpr (OP == 0) a[i] = fmaf(a[i], 1.0001f, 0.5f);                                  // v_fma_f32
            if constexpr (OP == 1) p[i] = __builtin_elementwise_fma(p[i], f32x2{1.0001f, 1.0001f}, f32x2{0.5f, 0.5f});   // v_pk_fma_f32
            if constexpr (OP == 2) { bf16x2 h = __builtin_convertvector(p[i], bf16x2); p[i].x += (float)h[0]; }          // cvt_pk + lshl + add
            if constexpr (OP == 3) u[i] = (u[i] & 0xffff0000u) + 3u;                                  // v_and + v_add
            if constexpr (OP == 4) a[i] = fmaxf(a[i], 0.25f) + 1.f;                                   // v_max + v_add
            if constexpr (OP == 5) p[i] = p[i] * f32x2{1.0001f, 1.0001f};                             // v_pk_mul_f32
        }
        if constexpr (OP == 6) {                                                                      // 2 independent MFMA chains
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m0, m1, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m1, m0, acc[1], 0, 0, 0);
        }
        if constexpr (OP == 7) {                                                                      // 1 dependent MFMA chain
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m0, m1, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m1, m0, acc[0], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < N; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
    s += acc[0][0] + acc[1][3];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 12345.678f) out[1] = 1;
}
template <int OP> void run(const char* name, int per_iter, unsigned long long* d) {
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, d, 1.0f);
    unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.2f cycles per instruction-group (%d per iteration)\n", name, (double)h / IT / per_iter, per_iter);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    run<0>("v_fma_f32", N, d); run<1>("v_pk_fma_f32", N, d); run<2>("v_cvt_pk_bf16_f32 + v_lshl + v_add", N, d);
    run<3>("v_and + v_add", N, d); run<4>("v_max + v_add", N, d); run<5>("v_pk_mul_f32", N, d);
    run<6>("mfma 32x32x16 bf16, 2 independent chains", 2, d); run<7>("mfma 32x32x16 bf16, dependent", 2, d);
    return 0;
}
