// Specialised finalize kernels (compute_global_heat_map, reference daam/trace.py:112-126) for the
// map sizes real pipelines produce with a 64x64 target:
//   * finalize_same_kernel : side == 64   -> clamp + mean only (pure streaming)
//   * finalize_up_kernel   : side 32 / 16 -> bicubic x2 / x4 + clamp + mean
// Everything else (x0.5 of SDXL-2048, 96x96 targets, odd sizes) takes the general kernel in
// daam_kernels.hip.  All of them add their share of the mean into `out` with f32 atomics.
//
// finalize_up_kernel: one WAVE walks a strided list of keys for one token.  Lane = output
// column; the lane keeps its whole output column (64 rows) in registers across all its keys.
// Per key: the [S,S] plane (2-4 KiB) is fetched with 16-byte loads one key ahead, widened to f32
// into a wave-private LDS tile, each lane gathers its 4 border-clamped x taps per source row
// (x pass -> S registers), then the y pass runs on registers with compile-time row taps; weights
// come from the host tables (bit-identical to torch's f32 coefficient arithmetic).
#include "daam_types.h"

namespace daam {

typedef float float4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <typename T> struct Plane;
template <> struct Plane<_Float16> {
    static constexpr int kPerPiece = 8;
    using Piece = half8;
    static __device__ __forceinline__ void widen(const Piece& p, float* dst) {
        *reinterpret_cast<float4v*>(dst) = float4v{(float)p[0], (float)p[1], (float)p[2], (float)p[3]};
        *reinterpret_cast<float4v*>(dst + 4) = float4v{(float)p[4], (float)p[5], (float)p[6], (float)p[7]};
    }
    static __device__ __forceinline__ void clamp_add(const Piece& p, float* a) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += fmaxf((float)p[i], 0.f);
    }
};
template <> struct Plane<float> {
    static constexpr int kPerPiece = 4;
    using Piece = float4v;
    static __device__ __forceinline__ void widen(const Piece& p, float* dst) { *reinterpret_cast<float4v*>(dst) = p; }
    static __device__ __forceinline__ void clamp_add(const Piece& p, float* a) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] += fmaxf(p[i], 0.f);
    }
};

// floor(((2o+1)*S - O) / (2*O)) = floor(src) of torch's area_pixel_compute_source_index (cubic)
template <int S, int O> __device__ __forceinline__ constexpr int src_floor(int o) {
    const int num = (2 * o + 1) * S - O, den = 2 * O;
    return num >= 0 ? num / den : -((-num + den - 1) / den);
}
template <int S> __device__ __forceinline__ constexpr int clamp_row(int v) { return v < 0 ? 0 : (v > S - 1 ? S - 1 : v); }

template <typename ACC_T, int S>
__global__ __launch_bounds__(256) void finalize_up_kernel(const FinLaunch L)
{
    constexpr int O = 64;
    constexpr int R = O / S;                                  // 2 or 4: weights repeat with period R
    using P = Plane<ACC_T>;
    constexpr int NP = S * S / P::kPerPiece;                  // 16-byte pieces per plane
    constexpr int PL = (NP + 63) / 64;                        // pieces per lane

    __shared__ __align__(16) float planes[4][S * S];
    __shared__ __align__(16) float red[O * O];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok = blockIdx.x;
    for (int i = tid; i < O * O; i += 256) red[i] = 0.f;

    const int tab = L.keys[0].tab;                            // one map size per launch
    const int16_t* tix = L.tab_idx + (size_t)tab * O * 4;
    const float* tw = L.tab_w + (size_t)tab * O * 4;
    const float* my = planes[wave];
    const float* x0 = my + tix[lane * 4 + 0];
    const float* x1 = my + tix[lane * 4 + 1];
    const float* x2 = my + tix[lane * 4 + 2];
    const float* x3 = my + tix[lane * 4 + 3];
    const float wx0 = tw[lane * 4 + 0], wx1 = tw[lane * 4 + 1], wx2 = tw[lane * 4 + 2], wx3 = tw[lane * 4 + 3];

    float acc[O];
#pragma unroll
    for (int i = 0; i < O; ++i) acc[i] = 0.f;

    const int stride = gridDim.y * 4;
    int kidx = blockIdx.y * 4 + wave;
    typename P::Piece pre[PL];
    auto fetch = [&](int k) {
        const ACC_T* src = reinterpret_cast<const ACC_T*>(L.keys[k].base) + (size_t)tok * S * S;
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            const int piece = lane + 64 * j;
            if (piece < NP) pre[j] = *as_global<typename P::Piece>(src + piece * P::kPerPiece);
        }
    };
    if (kidx < L.n_keys) fetch(kidx);
    for (; kidx < L.n_keys; kidx += stride) {
        float* mine = planes[wave];
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            const int piece = lane + 64 * j;
            if (piece < NP) P::widen(pre[j], mine + piece * P::kPerPiece);
        }
        if (kidx + stride < L.n_keys) fetch(kidx + stride);
        __builtin_amdgcn_wave_barrier();                       // wave-private tile: LDS ops of one wave stay in order
        float h[S];
#pragma unroll
        for (int y = 0; y < S; ++y)
            h[y] = x0[y * S] * wx0 + x1[y * S] * wx1 + x2[y * S] * wx2 + x3[y * S] * wx3;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int oy = 0; oy < O; ++oy) {
            constexpr int dummy = 0; (void)dummy;
            const int f = src_floor<S, O>(oy);
            const float* w = tw + (oy % R) * 4;               // uniform: scalar loads
            const float v = h[clamp_row<S>(f - 1)] * w[0] + h[clamp_row<S>(f)] * w[1] +
                            h[clamp_row<S>(f + 1)] * w[2] + h[clamp_row<S>(f + 2)] * w[3];
            acc[oy] += fmaxf(v, 0.f);
        }
    }
    __syncthreads();                                           // red[] zeroed
#pragma unroll
    for (int oy = 0; oy < O; ++oy) atomicAdd(&red[oy * O + lane], acc[oy]);     // ds_add_f32
    __syncthreads();
    float* out = L.out + (size_t)tok * O * O;
    for (int i = tid; i < O * O; i += 256) atomicAdd(out + i, red[i] * L.inv_n);
}

// side == out_side: out[t][i] += sum over this chunk's keys of max(plane[t][i], 0) / N
template <typename ACC_T>
__global__ __launch_bounds__(256) void finalize_same_kernel(const FinLaunch L)
{
    using P = Plane<ACC_T>;
    const int plane = L.out_side * L.out_side;
    const int vec = blockIdx.x * 256 + threadIdx.x;            // one 16-byte piece of one token plane
    const int per_tok = plane / P::kPerPiece;
    if (vec >= L.tokens * per_tok) return;
    const int tok = vec / per_tok, off = (vec - tok * per_tok) * P::kPerPiece;
    float a[P::kPerPiece];
#pragma unroll
    for (int i = 0; i < P::kPerPiece; ++i) a[i] = 0.f;
    int kidx = blockIdx.y;
    const int stride = gridDim.y;
#pragma unroll 4
    for (; kidx < L.n_keys; kidx += stride) {
        const ACC_T* src = reinterpret_cast<const ACC_T*>(L.keys[kidx].base) + (size_t)tok * plane + off;
        const typename P::Piece piece = *as_global<typename P::Piece>(src);
        P::clamp_add(piece, a);
    }
    float* out = L.out + (size_t)tok * plane + off;
#pragma unroll
    for (int i = 0; i < P::kPerPiece; ++i) atomicAdd(out + i, a[i] * L.inv_n);
}

// ---------------------------------------------------------------------------------------
hipError_t launch_finalize_same(const FinLaunch& L, int acc_dtype, hipStream_t stream, int* grid_out)
{
    const int plane = L.out_side * L.out_side;
    const int per = acc_dtype == 0 ? 8 : 4;
    const int vecs = L.tokens * plane / per;
    dim3 grid((vecs + 255) / 256, L.n_chunks);
    *grid_out = grid.x * grid.y;
    if (acc_dtype == 0) hipLaunchKernelGGL((finalize_same_kernel<_Float16>), grid, dim3(256), 0, stream, L);
    else hipLaunchKernelGGL((finalize_same_kernel<float>), grid, dim3(256), 0, stream, L);
    return hipGetLastError();
}

bool finalize_up_supported(int side, int out_side) { return out_side == 64 && (side == 32 || side == 16); }

hipError_t launch_finalize_up(const FinLaunch& L, int side, int acc_dtype, hipStream_t stream, int* grid_out)
{
    dim3 grid(L.tokens, L.n_chunks);
    *grid_out = grid.x * grid.y;
    if (side == 32) {
        if (acc_dtype == 0) hipLaunchKernelGGL((finalize_up_kernel<_Float16, 32>), grid, dim3(256), 0, stream, L);
        else hipLaunchKernelGGL((finalize_up_kernel<float, 32>), grid, dim3(256), 0, stream, L);
    } else {
        if (acc_dtype == 0) hipLaunchKernelGGL((finalize_up_kernel<_Float16, 16>), grid, dim3(256), 0, stream, L);
        else hipLaunchKernelGGL((finalize_up_kernel<float, 16>), grid, dim3(256), 0, stream, L);
    }
    return hipGetLastError();
}

}  // namespace daam
