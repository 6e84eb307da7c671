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
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

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
typedef unsigned short ushort8 __attribute__((ext_vector_type(8)));
template <> struct Plane<bf16_t> {                         // bf16 planes: widened by a shift
    static constexpr int kPerPiece = 8;
    using Piece = ushort8;
    static __device__ __forceinline__ float f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
    static __device__ __forceinline__ void widen(const Piece& p, float* dst) {
        *reinterpret_cast<float4v*>(dst) = float4v{f(p[0]), f(p[1]), f(p[2]), f(p[3])};
        *reinterpret_cast<float4v*>(dst + 4) = float4v{f(p[4]), f(p[5]), f(p[6]), f(p[7])};
    }
    static __device__ __forceinline__ void clamp_add(const Piece& p, float* a) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += fmaxf(f(p[i]), 0.f);
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
// x2 (32 -> 64) for fp16 planes with BOTH bicubic passes on the matrix cores.
//   pass 1 (x):  T = P Wx^T on v_mfma_f32_32x32x16_f16.  The A operand is the plane itself, 16 bytes per
//   lane straight from HBM (lane = source row y, half g -> columns 16ks + 8g .. +7); the B operand is the
//   banded 32x64 tap matrix (every tap weight of the x2 bicubic, and every border-merged sum of them, is
//   exactly representable in fp16 -- checked on the host, else the LDS kernel above is used).  Products
//   of two fp16 values are exact in f32 and at most 4 are non-zero per output, so T equals the
//   reference's f32 x interpolation up to the summation order.  C/D layout: lane (ox, g) holds rows
//   y = 8b + 4g + r of T[y][ox] in register 4b + r.
//   pass 2 (y):  out = Wy T  with T as the B operand.  The contraction index of an MFMA can be
//   permuted freely as long as A and B agree, so the C/D registers of pass 1 ARE a valid B operand:
//   k-step ks2, slot (g, i)  <->  y = 16 ks2 + 8 (i >> 2) + 4g + (i & 3) = the row held in register
//   8 ks2 + i.  The constant Wy pieces are built once per wave with the same permutation: no lane
//   exchange, no LDS.  T is fed as an fp16 pair hi + lo (hi = fp16(T), lo = fp16(T - hi)): 22
//   significant bits, |error| <= 2^-22 |T| -- below the f32 rounding noise of the four-tap sums it
//   feeds, far inside the tolerance of the bicubic parity tests; Wy is exact in fp16 (host check).
//   VALU work per plane: the hi/lo split + clamp / accumulate of the outputs (~165 instructions per 64x64
//   plane, against ~300 for a register y pass with lane exchanges); the MFMA pipe does 4 + 16 instructions.
//   Measured (SDXL-1024, 1000 planes x 77 tokens): ~68 us either way -- the kernel is bound by its
//   epilogue (4.1 M float atomics, ~24 us) and by HBM latency at 2 KB per plane, not by the loop.
//   C/D of pass 2: lane (ox, g) owns out[32mt + 8b + 4g + r][32nt + ox].
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ half2v fin_cvt_pk(float a, float b) {
    half2v r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// t - f32(h.lo) / t - f32(h.hi): one mixed-precision FMA straight from the packed fp16 pair
__device__ __forceinline__ float fin_sub_lo(float t, half2v h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(t));
    return r;
}
__device__ __forceinline__ float fin_sub_hi(float t, half2v h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(t));
    return r;
}
// max(a, b) for b >= 0 through the integer order of the bit patterns: one v_max_i32, no NaN canonicalisation in front
// (fmaxf() costs a second v_max_f32), and -- unlike an inline-asm v_max_f32 -- visible to the compiler's hazard
// recognizer, which pads the MFMA-result -> VALU-read wait states.  b >= 0: a negative a has its sign bit set and loses
// as an integer, two non-negative floats order like their bit patterns.
__device__ __forceinline__ float fin_max_nonneg(float a, float b) {
    const int x = __float_as_int(a), y = __float_as_int(b);
    return __int_as_float(x > y ? x : y);
}
// D = A B + C into NEW registers (C stays intact).  The builtin always comes out in the tied form (vdst = srcC) here, which
// costs a 16-register copy of the running sums in front of every chain; the three-address form does the "copy" in the
// matrix pipe.  Hazards of an MFMA hidden in an asm statement (cdna_hip_programming.md section 5.7): its A / B / C operands
// may have been written by the VALU instruction right before it (v_cvt_pk / v_max) -> s_nop 1 in front; its result is
// consumed only by the next MFMA of the chain, which takes it whole as C (no wait states needed); everything that READS
// an MFMA result with the VALU is compiler-generated code (no inline asm), so those wait states are padded for us.
__device__ __forceinline__ floatx16 fin_mfma_from(const half8& a, const half8& b, const floatx16& c) {
    floatx16 d;
    asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// ---------------------------------------------------------------------------------------
// x0.5 (128 -> 64: the 128x128 layers of SDXL at 2048 px, factor 0).  Pure streaming: a 32 KB plane is read once, row by row.
// One WAVE walks a strided list of keys for one token; lane = output column ox and reads the dword (two 16-bit elements) /
// the float2 that holds source columns 2ox, 2ox + 1 of every source row -- 256 contiguous bytes per wave and row; the outer taps
// 2ox - 1 and 2ox + 2 are the neighbour lanes' inner ones, fetched with one wave-shift DPP move each (the border lanes keep
// their own value = torch's index clamp).  A x0.5 bicubic has t = 0.5 everywhere: the same four weights (w0, w1, w1, w0) for
// every output, so  h = w1 (x0 + x1) + w0 (xl + xr)  per source row and  out = w1 (h[2oy] + h[2oy+1]) + w0 (h[2oy-1] + h[2oy+2])
// on a 4-deep window of row results; clamp, accumulate in the lane's 64 output registers.  Everything is unrolled at compile
// time (static register indices); rows are fetched 16 at a time, one batch ahead (across keys too).
// ---------------------------------------------------------------------------------------
template <typename T> struct Pair2;
template <> struct Pair2<_Float16> {
    using Raw = unsigned;
    static __device__ __forceinline__ void cvt(Raw r, float& a, float& b) {
        const half2v h = __builtin_bit_cast(half2v, r);
        a = (float)h[0]; b = (float)h[1];
    }
};
template <> struct Pair2<bf16_t> {
    using Raw = unsigned;
    static __device__ __forceinline__ void cvt(Raw r, float& a, float& b) { a = __uint_as_float(r << 16); b = __uint_as_float(r & 0xffff0000u); }
};
template <> struct Pair2<float> {
    using Raw = float2v;
    static __device__ __forceinline__ void cvt(Raw r, float& a, float& b) { a = r[0]; b = r[1]; }
};

template <typename ACC_T>
__global__ __launch_bounds__(256) void finalize_down2_kernel(const FinLaunch L)
{
    constexpr int O = 64, S = 128, RB = 16, NB = S / RB;
    using P2 = Pair2<ACC_T>;
    using Raw = typename P2::Raw;
    __shared__ __align__(16) float red[2 * O * O];
    constexpr int kMaxKeysPerWave = 64;
    __shared__ const void* kbase[4][kMaxKeysPerWave];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = blockIdx.x;
    const float* tw = L.tab_w + (size_t)L.keys[0].tab * O * 4;            // one map size per launch; t = 0.5 for every output
    const float w0 = tw[0], w1 = tw[1];
    // border clamp folded into the weights: lane 0's tap at column -1 is its own x0, lane 63's tap at column 128 its own x1
    const float w1a = lane == 0 ? w1 + w0 : w1, w1b = lane == 63 ? w1 + w0 : w1;

    float acc[O];
#pragma unroll
    for (int i = 0; i < O; ++i) acc[i] = 0.f;

    const int stride = gridDim.y * 4;
    const int first = blockIdx.y * 4 + wave;
    const int nk = first < L.n_keys ? min((L.n_keys - first + stride - 1) / stride, kMaxKeysPerWave) : 0;
    if (lane < nk) kbase[wave][lane] = as_global<FinKey>(L.keys)[first + lane * stride].base;
    __builtin_amdgcn_wave_barrier();

    auto row_ptr = [&](int ki) {
        return reinterpret_cast<const ACC_T*>(kbase[wave][ki]) + (size_t)tok * S * S + 2 * lane;
    };
    auto fetch = [&](const ACC_T* src, int b, Raw (&dst)[RB]) {
#pragma unroll
        for (int r = 0; r < RB; ++r) dst[r] = *as_global<Raw>(src + (size_t)(b * RB + r) * S);
    };
    if (nk > 0) {
        Raw buf[2][RB];
        fetch(row_ptr(0), 0, buf[0]);
        for (int ki = 0; ki < nk; ++ki) {
            const ACC_T* src = row_ptr(ki);
            const ACC_T* nxt = row_ptr(min(ki + 1, nk - 1));              // the last key re-reads its first batch (harmless)
            float hw[4] = {0.f, 0.f, 0.f, 0.f};                            // h[row - 3 .. row]
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b + 1 < NB) fetch(src, b + 1, buf[(b + 1) & 1]);
                else fetch(nxt, 0, buf[(b + 1) & 1]);
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const int row = b * RB + r;
                    float x0, x1;
                    P2::cvt(buf[b & 1][r], x0, x1);
                    // columns 2ox - 1 / 2ox + 2 = lane - 1's x1 / lane + 1's x0: DPP wave shift right / left by one lane with
                    // bound_ctrl (lanes 0 / 63 receive 0; their clamped border tap sits in w1a / w1b)
                    const float xl = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x1), 0x138, 0xf, 0xf, true));
                    const float xr = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x0), 0x130, 0xf, 0xf, true));
                    const float h = __builtin_fmaf(w1a, x0, __builtin_fmaf(w1b, x1, w0 * (xl + xr)));
                    hw[0] = hw[1]; hw[1] = hw[2]; hw[2] = hw[3]; hw[3] = h;
                    if (row == 0) hw[2] = h;                               // h[-1] := h[0]  (rows above the plane clamp to row 0)
                    if (row >= 2 && (row & 1) == 0) {                      // h[row - 3 .. row] = h[2oy - 1 .. 2oy + 2], oy = row / 2 - 1
                        const float v = __builtin_fmaf(w1, hw[1] + hw[2], w0 * (hw[0] + hw[3]));
                        acc[row / 2 - 1] += fmaxf(v, 0.f);
                    }
                }
            }
            // oy = 63: h[125], h[126], h[127], h[128] := h[127]
            const float v = __builtin_fmaf(w1, hw[2] + hw[3], w0 * (hw[1] + hw[3]));
            acc[O - 1] += fmaxf(v, 0.f);
        }
    }
    auto get = [&](int oy) -> float { return acc[oy]; };
    auto add = [&](int oy, float v) { acc[oy] += v; };
    wg_reduce_flush(red, wave, get, add, [&](int i) { return i * O + lane; }, L.out + (size_t)tok * O * O, L.inv_n);
}

// tunables of the x2 MFMA kernel: planes prefetched ahead per wave, waves per SIMD the register allocator must leave room for
constexpr int kFinKDepth = 2, kFinWaves = 4;

// body: workgroup (tok, chunk) of n_chunks key chunks
__device__ __forceinline__ void finalize_up32_mfma_body(const FinLaunch& L, const int tok, const int chunk, const int n_chunks)
{
    // wave = (nt, kq): output columns [32nt, 32nt+32) of every second key of the workgroup's chunk.  Half the
    // output per wave keeps the kernel under 128 VGPRs (4 waves per SIMD: the MFMA chain of one wave
    // runs under the VALU work of the others); the plane is fetched by both nt waves (second one hits L2).
    constexpr int S = 32, O = 64;
    constexpr int kDepth = kFinKDepth, kMaxKeysPerWave = 64;
    __shared__ const void* kbase[2][kMaxKeysPerWave];
    __shared__ __align__(16) float red[2][32 * 64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: loop bounds and branches stay scalar
    const int n = lane & 31, g = lane >> 5;
    const int nt = wave & 1, kq = wave >> 1;

    // operand pieces of the banded tap matrix, built on the host (build_up32_ops in daam_api.hip):
    //   wx[ks][e]    = W[32nt + n][16ks + 8g + e]                              (B of pass 1)
    //   wy[t][ks][i] = W[32t + n][16ks + 8(i >> 2) + 4g + (i & 3)]             (A of pass 2, permuted k)
    const DAAM_GLOBAL half8* ops = as_global<half8>(L.mfma_ops) + (size_t)(nt * 64 + lane) * 6;
    half8 wx[2], wy[2][2];
    wx[0] = ops[0]; wx[1] = ops[1];
    wy[0][0] = ops[2]; wy[0][1] = ops[3]; wy[1][0] = ops[4]; wy[1][1] = ops[5];

    floatx16 acc[2] = {floatx16{0}, floatx16{0}};             // [mt]

    // this chunk's key range (host-sized, see finalize_chunk_ranges); key lane kq takes every second key of it
    const int first = L.chunk_begin[chunk] + kq, end = L.chunk_begin[chunk + 1];
    constexpr int stride = 2;
    const int nk = first < end ? min((end - first + stride - 1) / stride, kMaxKeysPerWave) : 0;
    if (nt == 0 && lane < nk) kbase[kq][lane] = as_global<FinKey>(L.keys)[first + lane * stride].base;
    __syncthreads();

    half8 pre[kDepth][2];
    auto fetch = [&](int i, half8 (&dst)[2]) {
        const _Float16* src = reinterpret_cast<const _Float16*>(kbase[kq][i]) + (size_t)tok * S * S + n * S + 8 * g;
        dst[0] = *as_global<half8>(src);
        dst[1] = *as_global<half8>(src + 16);
    };
    // Straight-line pipeline: every iteration fetches unconditionally (indices past the end are clamped to the last key:
    // a harmless re-read), so the loop body has no control flow around its loads and the compiler can wait with COUNTED
    // vmcnt -- with the fetch inside an `if` it fell back to vmcnt(0), i.e. every iteration also waited for the plane it
    // had requested one iteration earlier and the prefetch distance collapsed to one.
    const int last = max(nk - 1, 0);
    if (nk > 0) {                                             // a wave without keys (fewer keys than key lanes) touches nothing
#pragma unroll
        for (int d = 0; d < kDepth; ++d) fetch(min(d, last), pre[d]);
    }

    // Issue budget per half plane (tools/ubench_issue.hip, gfx950): an MFMA 32x32x16 holds the matrix pipe for 32 cycles and
    // blocks VALU issue for ~10 of them; VALU of this and of the SIMD's other waves runs under the rest.  10 MFMAs =
    // 326 cycles of matrix pipe; the VALU side is 8 + 8 v_cvt_pk_f16_f32 and 16 v_fma_mix_f32 (5 cycles each) for the
    // hi / lo split and 32 v_max_i32 (4 cycles) = 288 cycles + 100 blocked: the two pipes are about balanced.
    //   * clamp + accumulate costs ONE v_max per output: pass 2 starts from the running sums (C = acc), so its result is
    //     D = acc + o and  acc + max(o, 0) == max(D, acc)  (fl(acc + o) >= acc exactly when o >= 0);
    //   * lo = T - hi is one mixed-precision FMA straight from the packed fp16 hi (written as T - (float)hi it becomes
    //     v_cvt_f32_f16 + v_sub_f32);
    //   * the max is v_max_i32 on the bit patterns (the sums are never negative): fmaxf() would add a canonicalising v_max.
    auto plane_step = [&](int ki, half8 (&p)[2], half8 (&p_next)[2]) {
        floatx16 c = __builtin_amdgcn_mfma_f32_32x32x16_f16(p[0], wx[0], floatx16{0}, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(p[1], wx[1], c, 0, 0, 0);
        fetch(min(ki + kDepth, last), p);
        half8 bhi[2], blo[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const float t0 = c[8 * ks + i], t1 = c[8 * ks + i + 1];
                // hi: compiler-generated v_cvt_pk_f16_f32 (RNE) -- the first VALU read of the MFMA result, padded by the
                // hazard recognizer; lo = T - hi: v_fma_mix_f32 through asm (left to itself the compiler vectorises the
                // pair into v_cvt_f32_f16 x2 + v_pk_fma_f32, which cannot run beside the matrix pipe); it depends on hi,
                // so it issues after that padded read
                const half2v hi = __builtin_convertvector(float2v{t0, t1}, half2v);
                const half2v lo = __builtin_convertvector(float2v{fin_sub_lo(t0, hi), fin_sub_hi(t1, hi)}, half2v);
                bhi[ks][i] = hi[0]; bhi[ks][i + 1] = hi[1];
                blo[ks][i] = lo[0]; blo[ks][i + 1] = lo[1];
            }
        floatx16 o0 = fin_mfma_from(wy[0][0], bhi[0], acc[0]);
        floatx16 o1 = fin_mfma_from(wy[1][0], bhi[0], acc[1]);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wy[0][1], bhi[1], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wy[1][1], bhi[1], o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wy[0][0], blo[0], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wy[1][0], blo[0], o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wy[0][1], blo[1], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wy[1][1], blo[1], o1, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            acc[0][v] = fin_max_nonneg(o0[v], acc[0][v]);
            acc[1][v] = fin_max_nonneg(o1[v], acc[1][v]);
        }
        // keep the steps of the unrolled body apart: the next step's plane becomes opaque HERE, so its pass 1 cannot be hoisted
        // to the top of the body (where it would have to wait for a plane fetched only one step earlier); at this point the
        // plane requested in this step may stay in flight (a counted vmcnt(2))
        asm volatile("" : "+v"(p_next[0]), "+v"(p_next[1]));
    };
    int ki = 0;
    for (; ki + kDepth <= nk; ki += kDepth) {                 // (never entered with nk == 0: `pre` is then unset and unused)
#pragma unroll
        for (int d = 0; d < kDepth; ++d) plane_step(ki + d, pre[d], pre[(d + 1) % kDepth]);
    }
#pragma unroll
    for (int d = 0; d < kDepth - 1; ++d)                      // remainder (nk not a multiple of kDepth): slots 0 .. in order
        if (ki + d < nk) plane_step(ki + d, pre[d], pre[(d + 1) % kDepth]);
    // the two key halves of an nt tile meet in LDS; the kq = 0 wave adds the sum into the output
    if (kq == 1) {
#pragma unroll
        for (int i = 0; i < 32; ++i) red[nt][i * 64 + lane] = acc[i >> 4][i & 15];
    }
    __syncthreads();
    if (kq == 0) {
        // (a two-level reduction -- per-workgroup slots of a partial buffer + one combine pass instead of these atomics --
        // measured 6 us SLOWER: the atomics are not what bounds the kernel)
        float* out = L.out + (size_t)tok * O * O + 32 * nt + n;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int row = 32 * (i >> 4) + 8 * ((i & 15) >> 2) + 4 * g + (i & 3);
            atomicAdd(out + row * O, (acc[i >> 4][i & 15] + red[nt][i * 64 + lane]) * L.inv_n);
        }
    }
}

__global__ __launch_bounds__(256, kFinWaves) void finalize_up32_mfma_kernel(const FinLaunch L)
{
    finalize_up32_mfma_body(L, blockIdx.x, blockIdx.y, gridDim.y);
}


// side == out_side: out[t][i] += sum over this chunk's keys of max(plane[t][i], 0) / N.
// A wave owns 64 consecutive 16-byte pieces of one token plane; kBatch keys are in flight per
// lane; the partial sums are transposed through a wave-private LDS tile so that the final
// atomics are 256-byte coalesced rows instead of 64 scattered 32-byte sectors.
template <typename ACC_T>
__device__ __forceinline__ void finalize_same_body(const FinLaunch& L, const int bx, const int by, const int ny)
{
    using P = Plane<ACC_T>;
    constexpr int E = P::kPerPiece;
    constexpr int kBatch = 8;
    __shared__ float tile[4][64 * (E + 1)];
    const int plane = L.out_side * L.out_side;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_id = bx * 4 + wave;                         // wave -> 64 pieces
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
    const int stride = ny;
    for (int k0 = by; k0 < L.n_keys; k0 += stride * kBatch) {
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

template <typename ACC_T>
__global__ __launch_bounds__(256) void finalize_same_kernel(const FinLaunch L)
{
    finalize_same_body<ACC_T>(L, blockIdx.x, blockIdx.y, gridDim.y);
}

// SDXL-1024 in fp16 has exactly two key classes: 64x64 planes (same size: an HBM stream) and 32x32 planes (x2 on the
// matrix cores: issue-bound).  One launch runs both side by side -- workgroups [0, up_blocks) are the x2 body, the rest
// the same-size body -- instead of two launches whose tails / ramps add up (measured 68 + 19 us back to back).
struct FinPair {
    FinLaunch up, same;
    int32_t up_blocks, same_gx, same_gy;
};

__global__ __launch_bounds__(256, kFinWaves) void finalize_up32_same_kernel(const FinPair P)
{
    const int b = blockIdx.x;
    if (b < P.up_blocks) {
        finalize_up32_mfma_body(P.up, b % P.up.tokens, b / P.up.tokens, P.up.n_chunks);
    } else {
        const int r = b - P.up_blocks;
        finalize_same_body<_Float16>(P.same, r % P.same_gx, r / P.same_gx, P.same_gy);
    }
}

static void same_grid(const FinLaunch& L, int acc_dtype, int* gx, int* gy)
{
    const int plane = L.out_side * L.out_side;
    const int per = acc_dtype == 1 ? 4 : 8;
    // waves never straddle token planes: per-token piece count rounded up to whole waves
    const int waves = L.tokens * ((plane / per + 63) / 64);
    *gx = (waves + 3) / 4;
    *gy = L.n_chunks;
}

// both fp16 classes of an SDXL-1024 finalize in one launch (up: x2 MFMA class, same: same-size class)
hipError_t launch_finalize_up32_same(const FinLaunch& up, const FinLaunch& same, hipStream_t stream, int* grid_out)
{
    FinPair P;
    P.up = up;
    P.same = same;
    P.up_blocks = up.tokens * up.n_chunks;
    same_grid(same, 0, &P.same_gx, &P.same_gy);
    const int grid = P.up_blocks + P.same_gx * P.same_gy;
    *grid_out = grid;
    hipLaunchKernelGGL(finalize_up32_same_kernel, dim3(grid), dim3(256), 0, stream, P);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
hipError_t launch_finalize_same(const FinLaunch& L, int acc_dtype, hipStream_t stream, int* grid_out)
{
    const int plane = L.out_side * L.out_side;
    const int per = acc_dtype == 1 ? 4 : 8;
    // waves never straddle token planes: per-token piece count rounded up to whole waves
    const int waves = L.tokens * ((plane / per + 63) / 64);
    dim3 grid((waves + 3) / 4, L.n_chunks);
    *grid_out = grid.x * grid.y;
    if (acc_dtype == 0) hipLaunchKernelGGL((finalize_same_kernel<_Float16>), grid, dim3(256), 0, stream, L);
    else if (acc_dtype == 2) hipLaunchKernelGGL((finalize_same_kernel<bf16_t>), grid, dim3(256), 0, stream, L);
    else hipLaunchKernelGGL((finalize_same_kernel<float>), grid, dim3(256), 0, stream, L);
    return hipGetLastError();
}

bool finalize_up_supported(int side, int out_side) { return out_side == 64 && (side == 32 || side == 16); }
bool finalize_down2_supported(int side, int out_side) { return out_side == 64 && side == 128; }

hipError_t launch_finalize_down2(const FinLaunch& L, int acc_dtype, hipStream_t stream, int* grid_out)
{
    dim3 grid(L.tokens, L.n_chunks);
    *grid_out = grid.x * grid.y;
    if (acc_dtype == 0) hipLaunchKernelGGL((finalize_down2_kernel<_Float16>), grid, dim3(256), 0, stream, L);
    else if (acc_dtype == 2) hipLaunchKernelGGL((finalize_down2_kernel<bf16_t>), grid, dim3(256), 0, stream, L);
    else hipLaunchKernelGGL((finalize_down2_kernel<float>), grid, dim3(256), 0, stream, L);
    return hipGetLastError();
}

hipError_t launch_finalize_up(const FinLaunch& L, int side, int acc_dtype, int mfma_ok, hipStream_t stream, int* grid_out)
{
    dim3 grid(L.tokens, L.n_chunks);
    *grid_out = grid.x * grid.y;
    if (side == 32 && acc_dtype == 0 && mfma_ok && L.mfma_ops) {
        hipLaunchKernelGGL(finalize_up32_mfma_kernel, grid, dim3(256), 0, stream, L);
    } else if (side == 32) {
        if (acc_dtype == 0) hipLaunchKernelGGL((finalize_up_kernel<_Float16, 32>), grid, dim3(256), 0, stream, L);
        else if (acc_dtype == 2) hipLaunchKernelGGL((finalize_up_kernel<bf16_t, 32>), grid, dim3(256), 0, stream, L);
        else hipLaunchKernelGGL((finalize_up_kernel<float, 32>), grid, dim3(256), 0, stream, L);
    } else {
        if (acc_dtype == 0) hipLaunchKernelGGL((finalize_up_kernel<_Float16, 16>), grid, dim3(256), 0, stream, L);
        else if (acc_dtype == 2) hipLaunchKernelGGL((finalize_up_kernel<bf16_t, 16>), grid, dim3(256), 0, stream, L);
        else hipLaunchKernelGGL((finalize_up_kernel<float, 16>), grid, dim3(256), 0, stream, L);
    }
    return hipGetLastError();
}

}  // namespace daam
