// cycles per MFMA instruction (gfx950), N independent accumulator chains per wave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define ITERS 4096
template <int CH> __global__ __launch_bounds__(256) void k32(float* out, _Float16 seed) {
    half8 a, b; for (int i = 0; i < 8; ++i) { a[i] = seed; b[i] = seed + (_Float16)i; }
    floatx16 c[CH]; for (int j = 0; j < CH; ++j) c[j] = floatx16{0};
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int j = 0; j < CH; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
    float s = 0; for (int j = 0; j < CH; ++j) s += c[j][0] + c[j][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH> __global__ __launch_bounds__(256) void k16(float* out, _Float16 seed) {
    half8 a, b; for (int i = 0; i < 8; ++i) { a[i] = seed; b[i] = seed + (_Float16)i; }
    floatx4 c[CH]; for (int j = 0; j < CH; ++j) c[j] = floatx4{0, 0, 0, 0};
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int j = 0; j < CH; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[j], 0, 0, 0);
    float s = 0; for (int j = 0; j < CH; ++j) s += c[j][0] + c[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> static void run(const char* name, K kern, float* out, int waves, int ch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256 * waves), dim3(256), 0, 0, out, (_Float16)1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256 * waves), dim3(256), 0, 0, out, (_Float16)1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)ITERS * ch * waves;
    printf("%-22s chains=%d waves/SIMD=%d  %7.3f ms  %6.2f ns per MFMA per SIMD\n", name, ch, waves, ms, ms * 1e6 / insts);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run("mfma_32x32x16_f16", k32<1>, out, w, 1);
        run("mfma_32x32x16_f16", k32<3>, out, w, 3);
        run("mfma_16x16x32_f16", k16<1>, out, w, 1);
        run("mfma_16x16x32_f16", k16<2>, out, w, 2);
        run("mfma_16x16x32_f16", k16<10>, out, w, 10);
    }
    return 0;
}
