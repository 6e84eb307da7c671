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

// Sum the four waves' register-resident [64,64] partial maps and add the result (x 1/N) into
// `out` with one coalesced f32 atomic per element.  Plain LDS stores / loads in two rounds: LDS
// float atomics (ds_add_f32) measured ~1000 cycles per wave-instruction here and were 50% of the
// kernel.  get(i) / add(i, v) access element i (compile-time) of the caller's registers, off(i) is
// its offset in the row-major tile (consecutive lanes -> consecutive offsets).
template <typename Get, typename Add, typename Off>
__device__ __forceinline__ void wg_reduce_flush(float* tiles /* [2][64*64] */, int wave, Get get, Add add, Off off,
                                                float* out, float inv_n)
{
    constexpr int O = 64;
    if (wave >= 2) {
#pragma unroll
        for (int i = 0; i < O; ++i) tiles[(wave - 2) * O * O + off(i)] = get(i);
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int i = 0; i < O; ++i) add(i, tiles[wave * O * O + off(i)]);
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int i = 0; i < O; ++i) tiles[off(i)] = get(i);
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < O; ++i) atomicAdd(out + off(i), (get(i) + tiles[off(i)]) * inv_n);
    }
}

template <typename ACC_T, int S>
__global__ __launch_bounds__(256) void finalize_up_kernel(const FinLaunch L)
{
    constexpr int O = 64;
    constexpr int R = O / S;                                  // 2 or 4: weights repeat with period R
    using P = Plane<ACC_T>;
    constexpr int NP = S * S / P::kPerPiece;                  // 16-byte pieces per plane
    constexpr int PL = (NP + 63) / 64;                        // pieces per lane

    __shared__ __align__(16) float planes[4][S * S];
    __shared__ __align__(16) float red[2 * O * O];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok = blockIdx.x;

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
    auto get = [&](int oy) -> float {
        if (P0 == 1 && oy == 0) return edge[0];
        if (P0 == 1 && oy == O - 1) return edge[1];
        return acc2[(oy - P0) >> 1][(oy - P0) & 1];
    };
    auto add = [&](int oy, float v) {
        if (P0 == 1 && oy == 0) edge[0] += v;
        else if (P0 == 1 && oy == O - 1) edge[1] += v;
        else acc2[(oy - P0) >> 1][(oy - P0) & 1] += v;
    };
    wg_reduce_flush(red, wave, get, add,
                    [&](int i) { return i * O + lane; }, L.out + (size_t)tok * O * O, L.inv_n);
}

// ---------------------------------------------------------------------------------------
// x2 (32 -> 64) for fp16 planes, no LDS on the data path.
//   x pass = T[y][ox] = sum_x A[y][x] Wx[ox][x] on the matrix cores (v_mfma_f32_32x32x16_f16): the
//   A operand is the plane itself, 16 bytes per lane straight from HBM (lane = source row y, half
//   g -> columns 16ks+8g..+7); the B operand is the banded 32x64 tap matrix built once per wave from
//   the host tables (every tap weight of the x2 bicubic, and every border-merged sum of them, is
//   exactly representable in fp16 -- checked on the host, else the LDS kernel is used).  Products of
//   two fp16 values are exact in f32 and at most 4 are non-zero per output, so T equals the
//   reference's f32 x interpolation up to the summation order.
//   The C/D layout leaves lane (j, g) with column ox = j (+32 for the second tile) and rows
//   {8b+4g+r}: 8 v_permlane32_swap per tile regroup that into rows [16g, 16g+16) of the column,
//   2 more swaps fetch the two halo rows on each side from the partner lane (border lanes clamp),
//   and the y pass + clamp + accumulate run on registers with compile-time row taps; lane half g
//   owns output rows [32g, 32g+32) of its column.
// ---------------------------------------------------------------------------------------
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void swap32(float& x, float& y) {
    // x <- {x.lo, y.lo}, y <- {x.hi, y.hi}   (lo / hi = lanes 0-31 / 32-63)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]);
    y = __uint_as_float(r[1]);
}

__global__ __launch_bounds__(256) void finalize_up32_mfma_kernel(const FinLaunch L)
{
    constexpr int S = 32, O = 64;
    constexpr int kDepth = 4, kMaxKeysPerWave = 64;
    __shared__ const void* kbase[4][kMaxKeysPerWave];
    __shared__ __align__(16) float red[2 * O * O];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, g = lane >> 5;
    const int tok = blockIdx.x;

    const int tab = as_global<FinKey>(L.keys)[0].tab;
    const int16_t* tix = L.tab_idx + (size_t)tab * O * 4;
    const float* tw = L.tab_w + (size_t)tab * O * 4;

    // B operand: wb[nt][ks][e] = Wx[ox = 32nt + n][x = 16ks + 8g + e]
    half8 wb[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int ox = 32 * nt + n;
        int ix[4];
        float wv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { ix[a] = tix[ox * 4 + a]; wv[a] = tw[ox * 4 + a]; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int x = 16 * ks + 8 * g + e;
                float w = 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a) w += (ix[a] == x) ? wv[a] : 0.f;
                wb[nt][ks][e] = (_Float16)w;
            }
    }
    // y weights: output row parity 0 (t = 0.75) and 1 (t = 0.25)
    float wy[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int a = 0; a < 4; ++a) wy[p][a] = tw[p * 4 + a];

    // acc[nt][o'] = output row 32g + o' of column 32nt + n
    float acc[2][32];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[nt][i] = 0.f;

    const int stride = gridDim.y * 4;
    const int first = blockIdx.y * 4 + wave;
    const int nk = first < L.n_keys ? min((L.n_keys - first + stride - 1) / stride, kMaxKeysPerWave) : 0;
    if (lane < nk) kbase[wave][lane] = as_global<FinKey>(L.keys)[first + lane * stride].base;
    __builtin_amdgcn_wave_barrier();

    half8 pre[kDepth][2];
    auto fetch = [&](int i, half8 (&dst)[2]) {
        const _Float16* src = reinterpret_cast<const _Float16*>(kbase[wave][i]) + (size_t)tok * S * S + n * S + 8 * g;
        dst[0] = *as_global<half8>(src);
        dst[1] = *as_global<half8>(src + 16);
    };
#pragma unroll
    for (int d = 0; d < kDepth; ++d)
        if (d < nk) fetch(d, pre[d]);

    for (int i0 = 0; i0 < nk; i0 += kDepth) {
#pragma unroll
      for (int d = 0; d < kDepth; ++d) {
        const int ki = i0 + d;
        if (ki >= nk) break;
        floatx16 c[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            c[nt] = floatx16{0};
            c[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pre[d][0], wb[nt][0], c[nt], 0, 0, 0);
            c[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pre[d][1], wb[nt][1], c[nt], 0, 0, 0);
        }
        if (ki + kDepth < nk) fetch(ki + kDepth, pre[d]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            // e[q + 2] = source row 16g + q of this lane's column, q = -2 .. 17
            float e[20];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = c[nt][4 * b + r], y = c[nt][4 * (b + 2) + r];
                    swap32(x, y);
                    e[2 + 8 * b + r] = x;                       // row 16g + 8b + r
                    e[2 + 8 * b + 4 + r] = y;                   // row 16g + 8b + 4 + r
                }
            {
                float xa = e[2 + 0], ya = e[2 + 14], xb = e[2 + 1], yb = e[2 + 15];
                swap32(xa, ya);                                 // upper lanes: xa = partner's q=14; lower: ya = partner's q=0
                swap32(xb, yb);                                 // upper lanes: xb = partner's q=15; lower: yb = partner's q=1
                e[0] = g ? xa : e[2];                           // rows -2, -1 clamp to row 0 for the lower half
                e[1] = g ? xb : e[2];
                e[18] = g ? e[17] : ya;                         // rows 32, 33 clamp to row 31 for the upper half
                e[19] = g ? e[17] : yb;
            }
            // output row o' (global 32g + o'): taps local q = f'-1 .. f'+2, f' = floor(o'/2 - 1/4)
            {
                // singles o' = 0 (f' = -1) and o' = 31 (f' = 15)
                float v0 = e[0] * wy[0][0];
                v0 = __builtin_fmaf(e[1], wy[0][1], v0);
                v0 = __builtin_fmaf(e[2], wy[0][2], v0);
                v0 = __builtin_fmaf(e[3], wy[0][3], v0);
                acc[nt][0] += fmaxf(v0, 0.f);
                float v1 = e[16] * wy[1][0];
                v1 = __builtin_fmaf(e[17], wy[1][1], v1);
                v1 = __builtin_fmaf(e[18], wy[1][2], v1);
                v1 = __builtin_fmaf(e[19], wy[1][3], v1);
                acc[nt][31] += fmaxf(v1, 0.f);
            }
            // pairs (o', o'+1), o' odd: both have f' = (o'-1)/2, taps q = f'-1 .. f'+2
            constexpr int YB = 5;
#pragma unroll
            for (int p0 = 0; p0 < 15; p0 += YB) {
                float2v v[YB];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < YB; ++j) {
                        const int op = 1 + 2 * (p0 + j);                  // odd o'
                        const int f = (op - 1) / 2;
                        const float hv = e[2 + f - 1 + a];
                        const float2v w = {wy[1][a], wy[0][a]};          // (odd row, even row)
                        v[j] = a == 0 ? float2v{hv, hv} * w : __builtin_elementwise_fma(float2v{hv, hv}, w, v[j]);
                    }
#pragma unroll
                for (int j = 0; j < YB; ++j) {
                    const int op = 1 + 2 * (p0 + j);
                    acc[nt][op] += fmaxf(v[j][0], 0.f);
                    acc[nt][op + 1] += fmaxf(v[j][1], 0.f);
                }
            }
        }
      }
    }
    // element i = (tile i / 32, local row i % 32)  ->  out[32g + i % 32][32 (i / 32) + n]
    wg_reduce_flush(red, wave, [&](int i) { return acc[i >> 5][i & 31]; },
                    [&](int i, float v) { acc[i >> 5][i & 31] += v; },
                    [&](int i) { return (32 * g + (i & 31)) * O + 32 * (i >> 5) + n; },
                    L.out + (size_t)tok * O * O, L.inv_n);
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

hipError_t launch_finalize_up(const FinLaunch& L, int side, int acc_dtype, int mfma_ok, hipStream_t stream, int* grid_out)
{
    dim3 grid(L.tokens, L.n_chunks);
    *grid_out = grid.x * grid.y;
    if (side == 32 && acc_dtype == 0 && mfma_ok) {
        hipLaunchKernelGGL(finalize_up32_mfma_kernel, grid, dim3(256), 0, stream, L);
    } else if (side == 32) {
        if (acc_dtype == 0) hipLaunchKernelGGL((finalize_up_kernel<_Float16, 32>), grid, dim3(256), 0, stream, L);
        else hipLaunchKernelGGL((finalize_up_kernel<float, 32>), grid, dim3(256), 0, stream, L);
    } else {
        if (acc_dtype == 0) hipLaunchKernelGGL((finalize_up_kernel<_Float16, 16>), grid, dim3(256), 0, stream, L);
        else hipLaunchKernelGGL((finalize_up_kernel<float, 16>), grid, dim3(256), 0, stream, L);
    }
    return hipGetLastError();
}

}  // namespace daam
