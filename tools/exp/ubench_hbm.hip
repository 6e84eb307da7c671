// HBM read-bandwidth probes on gfx950 (tools only): what can a read-dominated kernel reach, and how
// much does the tap kernel's access pattern (64-byte pieces of 128-byte rows at a 2560-byte stride,
// one tensor per step) cost against a plain stream?
//   hipcc --offload-arch=gfx950 -O3 tools/exp/ubench_hbm.hip -o tools/exp/ubench_hbm && tools/exp/ubench_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float float4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// plain stream: every lane 16 bytes, consecutive lanes consecutive addresses, UNROLL loads in flight
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_read(const float4v* __restrict__ src, size_t n16, float* sink) {
    float4v acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        float4v v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

// tap-like: workgroup = (head, 128-pixel tile) of a [pixels][heads*64] fp16 tensor; lane (pixel j, quarter h)
// reads 16 B at +16h and +64+16h of its pixel's 128-byte head row; `steps` tensors in sequence.
__global__ __launch_bounds__(256) void tap_like(const char* __restrict__ base, size_t tensor_bytes, int steps, int heads,
                                                int tiles, float* sink, int head_major) {
    const int wg = blockIdx.x;
    const int head = wg / tiles, tile = wg - head * tiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, h = lane >> 4;
    float4v acc = {0, 0, 0, 0};
    const size_t row = (size_t)heads * 128;
    const size_t hw = (size_t)tiles * 128;
    for (int s = 0; s < steps; ++s) {
        const char* t = base + (size_t)s * tensor_bytes;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const size_t px = (size_t)tile * 128 + wave * 32 + g * 16 + j;
            const char* p = head_major ? t + ((size_t)head * hw + px) * 128 + h * 16 : t + px * row + head * 128 + h * 16;
            acc += *reinterpret_cast<const float4v*>(p);
            acc += *reinterpret_cast<const float4v*>(p + 64);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

int main() {
    const size_t bytes = (size_t)10 << 30;
    char* buf;
    float* sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](auto&& launch, double gb, const char* name) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms / 5, gb / (ms / 5 * 1e-3));
    };
    const size_t n16 = bytes / 16;
    for (int grid : {1024, 2048, 4096, 16384})
        for (int un : {4, 8}) {
            char name[96];
            snprintf(name, sizeof name, "stream grid=%d unroll=%d", grid, un);
            if (un == 4) time([&] { hipLaunchKernelGGL(stream_read<4>, dim3(grid), dim3(256), 0, 0, (const float4v*)buf, n16, sink); }, bytes / 1e9, name);
            else time([&] { hipLaunchKernelGGL(stream_read<8>, dim3(grid), dim3(256), 0, 0, (const float4v*)buf, n16, sink); }, bytes / 1e9, name);
        }
    // tap-like: 20 heads x 1024 pixels per tensor (2.6 MB read per tensor per step), 64 such "layers" side by side
    // would need a table; instead make the tensor big: 20 heads, tiles = 512 (65536 pixels): 168 MB per step, 50 steps = 8.4 GB
    const int heads = 20, tiles = 512, steps = 50;
    const size_t tensor = (size_t)heads * 128 * tiles * 128;
    for (int hm = 0; hm < 2; ++hm) {
        char name[96];
        snprintf(name, sizeof name, "tap-like 20 heads, %d steps, head_major=%d", steps, hm);
        time([&] { hipLaunchKernelGGL(tap_like, dim3(heads * tiles), dim3(256), 0, 0, buf, tensor, steps, heads, tiles, sink, hm); },
             (double)tensor * steps / 1e9, name);
    }
    return 0;
}
