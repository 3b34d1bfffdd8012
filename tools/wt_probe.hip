// Developer probe (round 5): when may a tensor be written with sc1 (write-through) stores and read by the NEXT kernel with plain loads?
// R: every block reads the buffer (plain loads, warms its CU's L1 / its XCD's L2) and writes a checksum; W: rewrites the buffer with value k by
// plain | nt | sc1 stores, from the SAME or from OTHER blocks than those that read it; R again: does it see k everywhere?
//   hipcc --offload-arch=gfx950 -O3 tools/wt_probe.hip -o tools/wt_probe && tools/wt_probe
// (Round 6: the one failure that motivated this probe -- the Adam sweep with sc1 stores, DESIGN 5.000 -- was NOT a visibility effect, which is why none
// of these 72 patterns shows it: an asm-issued 16-byte store followed at once by a VALU write of its data registers, a wait-state hazard hipcc pads only
// for its own stores.  tools/sc1_adam_repro.py reproduces and bisects it; the conv epilogue's stores carry `s_nop 1` since.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f4;
__global__ void rd(const float* __restrict__ buf, int n_per_block, float* out, int rot) {
    const int b = (blockIdx.x + rot) % gridDim.x;
    const f4* p = reinterpret_cast<const f4*>(buf + (size_t)b * n_per_block);
    float s = 0.f, mn = 1e30f, mx = -1e30f;
    for (int i = threadIdx.x; i < n_per_block / 4; i += blockDim.x) { const f4 v = p[i]; for (int j = 0; j < 4; ++j) { s += v[j]; mn = fminf(mn, v[j]); mx = fmaxf(mx, v[j]); } }
    __shared__ float smn[256], smx[256];
    smn[threadIdx.x] = mn; smx[threadIdx.x] = mx; __syncthreads();
    if (threadIdx.x == 0) { for (int i = 1; i < blockDim.x; ++i) { mn = fminf(mn, smn[i]); mx = fmaxf(mx, smx[i]); } out[2 * b] = mn; out[2 * b + 1] = mx; }
}
template <int MODE> __global__ void wr(float* __restrict__ buf, int n_per_block, float val, int rot, int rmw) {
    const int b = (blockIdx.x + rot) % gridDim.x;
    f4* p = reinterpret_cast<f4*>(buf + (size_t)b * n_per_block);
    for (int i = threadIdx.x; i < n_per_block / 4; i += blockDim.x) {
        f4 v = {val, val, val, val};
        if (rmw) { const f4 o = p[i]; v += (o - o); }                        // read-modify-write: the line is in this CU's L1 when the store leaves
        if (MODE == 0) p[i] = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, p + i);
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p + i), "v"(v) : "memory");
    }
}
int main() {
    const int NB = 1024, NT = 256;
    const char* names[3] = {"plain", "nt", "sc1"};
    for (int npb : {1024, 16384, 262144}) {
        float *buf, *out; hipMalloc(&buf, (size_t)NB * npb * 4); hipMalloc(&out, NB * 2 * 4);
        std::vector<float> h(NB * 2);
        for (int mode = 0; mode < 3; ++mode)
            for (int rmw = 0; rmw < 2; ++rmw)
                for (int rotw : {0, 1, 8, 129}) {                            // writer block b writes the slice reader block b + rotw reads: 0 = same block index (same XCD, likely same CU), 1 = next XCD, 8 = same XCD other CU
                    int bad = 0;
                    for (int k = 1; k <= 200; ++k) {
                        hipLaunchKernelGGL(rd, dim3(NB), dim3(NT), 0, 0, buf, npb, out, 0);
                        if (mode == 0) hipLaunchKernelGGL(wr<0>, dim3(NB), dim3(NT), 0, 0, buf, npb, (float)k, rotw, rmw);
                        else if (mode == 1) hipLaunchKernelGGL(wr<1>, dim3(NB), dim3(NT), 0, 0, buf, npb, (float)k, rotw, rmw);
                        else hipLaunchKernelGGL(wr<2>, dim3(NB), dim3(NT), 0, 0, buf, npb, (float)k, rotw, rmw);
                        hipLaunchKernelGGL(rd, dim3(NB), dim3(NT), 0, 0, buf, npb, out, 0);
                        hipMemcpy(h.data(), out, NB * 2 * 4, hipMemcpyDeviceToHost);
                        for (int b = 0; b < NB; ++b) if (h[2 * b] != (float)k || h[2 * b + 1] != (float)k) { ++bad; break; }
                    }
                    printf("bytes/block %7d  store %-5s rmw %d writer-rotation %3d: %d of 200 rounds saw stale data\n", npb * 4, names[mode], rmw, rotw, bad);
                }
        hipFree(buf); hipFree(out);
    }
    return 0;
}
