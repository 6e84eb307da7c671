// Do two small kernels on two non-blocking streams overlap on this box?  (tools only)
//   hipcc --offload-arch=gfx950 -O3 tools/exp/ubench_streams.hip -o tools/exp/ubench_streams && tools/exp/ubench_streams
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void spin(float* sink, int iters) {
    float x = threadIdx.x * 0.5f;
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    if (x == 1234.5f) *sink = x;
}

int main() {
    float* sink;
    CK(hipMalloc(&sink, 4));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const int iters = 100000;                                   // ~200 us of dependent FMAs
    auto wall = [&](auto&& fn) {
        fn(); CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 10; ++r) fn();
        CK(hipDeviceSynchronize());
        printf("%8.1f us per round\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10);
        return 0;
    };
    for (int grid : {48, 256, 1024, 2048}) {
        printf("grid %4d  one stream, two launches:      ", grid);
        wall([&] { hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, a, sink, iters); hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, a, sink, iters); });
        printf("grid %4d  two independent streams:        ", grid);
        wall([&] { hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, a, sink, iters); hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, b, sink, iters); });
        printf("grid %4d  fork / join by events (a -> b): ", grid);
        wall([&] {
            hipEventRecord(fork, a); hipStreamWaitEvent(b, fork, 0);
            hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, a, sink, iters);
            hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, b, sink, iters);
            hipEventRecord(join, b); hipStreamWaitEvent(a, join, 0);
        });
    }
    return 0;
}
