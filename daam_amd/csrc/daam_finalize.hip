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
typedef float float2v __attribute__((ext_vector_type(2)));
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

    // this lane's output column: acc2[i] = rows (P0 + 2i, P0 + 2i + 1); with P0 = 1 rows 0 and 63 in edge[]
    constexpr int P0 = (R == 2) ? 1 : 0;
    static_assert(src_floor<S, O>(P0) == src_floor<S, O>(P0 + 1), "paired output rows must share their taps");
    float2v acc2[O / 2];
    float edge[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < O / 2; ++i) acc2[i] = float2v{0.f, 0.f};

    // this wave's keys: first, first + stride, ...   Their plane base pointers go to LDS once
    // (no dependent table fetch per key), and the planes are fetched kDepth keys ahead: a wave's
    // critical path is then one HBM latency per kDepth planes instead of two per plane.
    constexpr int kDepth = 4;
    constexpr int kMaxKeysPerWave = 64;
    __shared__ const void* kbase[4][kMaxKeysPerWave];
    const int stride = gridDim.y * 4;
    const int first = blockIdx.y * 4 + wave;
    const int nk = first < L.n_keys ? min((L.n_keys - first + stride - 1) / stride, kMaxKeysPerWave) : 0;
    if (lane < nk) kbase[wave][lane] = as_global<FinKey>(L.keys)[first + lane * stride].base;
    __builtin_amdgcn_wave_barrier();
    typename P::Piece pre[kDepth][PL];
    auto fetch = [&](int i, typename P::Piece (&dst)[PL]) {
        const ACC_T* src = reinterpret_cast<const ACC_T*>(kbase[wave][i]) + (size_t)tok * S * S;
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            const int piece = lane + 64 * j;
            if (piece < NP) dst[j] = *as_global<typename P::Piece>(src + piece * P::kPerPiece);
        }
    };
#pragma unroll
    for (int d = 0; d < kDepth; ++d)
        if (d < nk) fetch(d, pre[d]);
    for (int i0 = 0; i0 < nk; i0 += kDepth) {
#pragma unroll
      for (int d = 0; d < kDepth; ++d) {
        const int ki = i0 + d;
        if (ki >= nk) break;
        float* mine = planes[wave];
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            const int piece = lane + 64 * j;
            if (piece < NP) P::widen(pre[d][j], mine + piece * P::kPerPiece);
        }
        if (ki + kDepth < nk) fetch(ki + kDepth, pre[d]);
        __builtin_amdgcn_wave_barrier();                       // wave-private tile: LDS ops of one wave stay in order
        // x pass on row PAIRS (v_pk_fma_f32: two rows per instruction); h2[yp] = (h[2yp], h[2yp+1]).
        // LDS gathers are issued XB row pairs ahead of their use so their latency overlaps.
        float2v h2[S / 2];
        constexpr int XB = 4;
#pragma unroll
        for (int y0 = 0; y0 < S / 2; y0 += XB) {
            float2v t0[XB], t1[XB], t2[XB], t3[XB];
#pragma unroll
            for (int j = 0; j < XB; ++j) {
                const int ra = 2 * (y0 + j) * S, rb = ra + S;
                t0[j] = float2v{x0[ra], x0[rb]};
                t1[j] = float2v{x1[ra], x1[rb]};
                t2[j] = float2v{x2[ra], x2[rb]};
                t3[j] = float2v{x3[ra], x3[rb]};
            }
            __builtin_amdgcn_sched_barrier(0);
            // tap-major: XB independent accumulation chains in flight (a dependent v_pk_fma_f32 cannot
            // issue back-to-back)
            float2v v[XB];
#pragma unroll
            for (int j = 0; j < XB; ++j) v[j] = t0[j] * wx0;
#pragma unroll
            for (int j = 0; j < XB; ++j) v[j] = __builtin_elementwise_fma(t1[j], float2v{wx1, wx1}, v[j]);
#pragma unroll
            for (int j = 0; j < XB; ++j) v[j] = __builtin_elementwise_fma(t2[j], float2v{wx2, wx2}, v[j]);
#pragma unroll
            for (int j = 0; j < XB; ++j) h2[y0 + j] = __builtin_elementwise_fma(t3[j], float2v{wx3, wx3}, v[j]);
        }
        __builtin_amdgcn_wave_barrier();
        // y pass on OUTPUT row pairs that share their 4 source rows (R = 2: (1,2), (3,4), ..., rows 0
        // and 63 alone; R = 4: (0,1), (2,3), ...): each source row is broadcast against the pair of
        // its two coefficients.
        auto hrow = [&](int r) { const int rc = clamp_row<S>(r); return h2[rc >> 1][rc & 1]; };
        if (P0 == 1) {
            const int fa = src_floor<S, O>(0), fb = src_floor<S, O>(O - 1);
            const float* wa = tw + (0 % R) * 4;
            const float* wb = tw + ((O - 1) % R) * 4;
            float va = hrow(fa - 1) * wa[0], vb = hrow(fb - 1) * wb[0];
#pragma unroll
            for (int a = 1; a < 4; ++a) {
                va = __builtin_fmaf(hrow(fa - 1 + a), wa[a], va);
                vb = __builtin_fmaf(hrow(fb - 1 + a), wb[a], vb);
            }
            edge[0] += fmaxf(va, 0.f);
            edge[1] += fmaxf(vb, 0.f);
        }
        constexpr int YB = 8;                                  // independent output pairs in flight
        constexpr int NPAIR = (O - P0) / 2;
#pragma unroll
        for (int p0 = 0; p0 < NPAIR; p0 += YB) {
            float2v v[YB];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int j = 0; j < YB; ++j) {
                    const int p = p0 + j;
                    if (p < NPAIR) {
                        const int o = P0 + 2 * p;
                        const int f = src_floor<S, O>(o);      // == src_floor(o + 1) by construction
                        const float hv = hrow(f - 1 + a);
                        const float2v w = {tw[(o % R) * 4 + a], tw[((o + 1) % R) * 4 + a]};   // uniform: scalar loads
                        v[j] = a == 0 ? float2v{hv, hv} * w : __builtin_elementwise_fma(float2v{hv, hv}, w, v[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < YB; ++j)
                if (p0 + j < NPAIR) acc2[p0 + j] += float2v{fmaxf(v[j][0], 0.f), fmaxf(v[j][1], 0.f)};
        }
      }
    }
    __syncthreads();                                           // red[] zeroed
#pragma unroll
    for (int oy = 0; oy < O; ++oy) {
        float v;
        if (P0 == 1 && oy == 0) v = edge[0];
        else if (P0 == 1 && oy == O - 1) v = edge[1];
        else v = acc2[(oy - P0) >> 1][(oy - P0) & 1];
        atomicAdd(&red[oy * O + lane], v);                     // ds_add_f32
    }
    __syncthreads();
    float* out = L.out + (size_t)tok * O * O;
    for (int i = tid; i < O * O; i += 256) atomicAdd(out + i, red[i] * L.inv_n);
}

// side == out_side: out[t][i] += sum over this chunk's keys of max(plane[t][i], 0) / N.
// A wave owns 64 consecutive 16-byte pieces of one token plane; kBatch keys are in flight per
// lane; the partial sums are transposed through a wave-private LDS tile so that the final
// atomics are 256-byte coalesced rows instead of 64 scattered 32-byte sectors.
template <typename ACC_T>
__global__ __launch_bounds__(256) void finalize_same_kernel(const FinLaunch L)
{
    using P = Plane<ACC_T>;
    constexpr int E = P::kPerPiece;
    constexpr int kBatch = 8;
    __shared__ float tile[4][64 * (E + 1)];
    const int plane = L.out_side * L.out_side;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_id = blockIdx.x * 4 + wave;                 // wave -> 64 pieces
    const int per_tok = plane / E;                             // pieces per token plane
    const int waves_per_tok = (per_tok + 63) / 64;             // a wave never straddles two token planes
    const int tok = wave_id / waves_per_tok;
    if (tok >= L.tokens) return;
    const int piece0 = (wave_id - tok * waves_per_tok) * 64;
    const int off = (piece0 + lane) * E;
    const bool valid = piece0 + lane < per_tok;
    float a[E];
#pragma unroll
    for (int i = 0; i < E; ++i) a[i] = 0.f;
    const int stride = gridDim.y;
    for (int k0 = blockIdx.y; k0 < L.n_keys; k0 += stride * kBatch) {
        typename P::Piece buf[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int kk = min(k0 + j * stride, L.n_keys - 1);   // clamped duplicate, masked below
            const ACC_T* src = reinterpret_cast<const ACC_T*>(L.keys[kk].base) + (size_t)tok * plane + off;
            if (valid) buf[j] = *as_global<typename P::Piece>(src);
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j)
            if (valid && k0 + j * stride < L.n_keys) P::clamp_add(buf[j], a);
    }
    float* t = tile[wave];
#pragma unroll
    for (int i = 0; i < E; ++i) t[lane * (E + 1) + i] = a[i] * L.inv_n;
    __builtin_amdgcn_wave_barrier();
    float* out = L.out + (size_t)tok * plane + piece0 * E;
    const int span = min(64, per_tok - piece0) * E;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int e = i * 64 + lane;                           // element of the wave's contiguous span
        if (e < span) atomicAdd(out + e, t[(e / E) * (E + 1) + (e % E)]);
    }
}

// ---------------------------------------------------------------------------------------
hipError_t launch_finalize_same(const FinLaunch& L, int acc_dtype, hipStream_t stream, int* grid_out)
{
    const int plane = L.out_side * L.out_side;
    const int per = acc_dtype == 0 ? 8 : 4;
    // waves never straddle token planes: per-token piece count rounded up to whole waves
    const int waves = L.tokens * ((plane / per + 63) / 64);
    dim3 grid((waves + 3) / 4, L.n_chunks);
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
