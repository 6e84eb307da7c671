// Do MFMA and VALU work of DIFFERENT waves on one SIMD overlap on gfx950?  (tools only)
//   hipcc --offload-arch=gfx950 -O3 tools/exp/ubench_overlap.hip -o tools/exp/ubench_overlap && tools/exp/ubench_overlap
// One workgroup of 8 waves per CU (waves w and w+4 share SIMD w): mode 0 = waves 0-3 run an MFMA loop, 4-7 idle;
// mode 1 = waves 4-7 run a VALU loop, 0-3 idle; mode 2 = both.  Overlap => t(2) ~ max(t0, t1); none => t0 + t1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(512) void k(int mode, int iters, float* sink) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = wave < 4 && (mode == 0 || mode == 2);
    const bool do_valu = wave >= 4 && (mode == 1 || mode == 2);
    if (do_mfma) {
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
        floatx4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
        }
        if (c0[0] + c1[1] + c2[2] + c3[3] == 1234.5f) *sink = 1.f;
    }
    if (mode == 3 && wave < 4) {                              // ONE wave issues both, interleaved by the compiler
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
        floatx4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        float x0 = threadIdx.x * 0.25f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f);
            x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f);
            x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
            x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f);
            x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
            x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f);
            x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
        }
        if (c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 == 1234.5f) *sink = 3.f;
    }
    if (do_valu) {
        float x0 = threadIdx.x * 0.25f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {                      // 16 independent FMAs per iteration
                x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f);
                x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
            }
        }
        if (x0 + x1 + x2 + x3 == 1234.5f) *sink = 2.f;
    }
}

int main() {
    float* sink;
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200000;
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, 1000, sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d (%s): %.3f ms  -> %.1f ns per iteration (4 MFMA 16x16x32 and/or 16 v_fma)\n", mode,
               mode == 0 ? "MFMA waves only" : mode == 1 ? "VALU waves only" : mode == 2 ? "both, different waves" : "both, same wave", ms, ms * 1e6 / iters);
    }
    return 0;
}
