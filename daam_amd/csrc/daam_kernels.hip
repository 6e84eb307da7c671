// Baseline (any head_dim, any dtype combo) kernels of the DAAM heat-map extraction path for
// gfx950.  The fp16 MFMA tap lives in daam_tap_mfma.hip; this file holds
//   * tap_generic_kernel   : softmax(scale q k^T) -> conditional half -> running sums
//                            (reference daam/trace.py:276-294 + daam/heatmap.py:153-156)
//   * tap_probs_kernel     : the same accumulate from materialised probabilities
//   * finalize_kernel      : bicubic -> clamp -> mean over keys (daam/trace.py:112-126)
//   * normalize_kernel     : daam/trace.py:129-130
//   * word map kernels     : daam/heatmap.py:121-123, 77-93
#include "daam_types.h"

namespace daam {

// ---------------------------------------------------------------------------------------
// dtype helpers.  "round_to<T>" reproduces the rounding point of a tensor that the reference
// pipeline materialises in dtype T (fp16 logits / probabilities), returning the value as f32.
// ---------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<__half>(const __half* p) { return __half2float(*p); }
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

template <typename T> __device__ __forceinline__ float round_to(float x);
template <> __device__ __forceinline__ float round_to<__half>(float x) { return __half2float(__float2half_rn(x)); }
template <> __device__ __forceinline__ float round_to<float>(float x) { return x; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float x) { return bf16_to_f32(f32_to_bf16(x)); }

// acc = acc + x in the accumulator dtype.  For fp16 the f32 add of two fp16 values followed by
// one RNE rounding is the correctly rounded fp16 add (24 >= 2*11+2 bits), i.e. exactly what
// torch's / numpy's half add does (heatmap.py:156).
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<__half>(__half* p, float v) { *p = __float2half_rn(v); }
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
// bf16: f32 add of two bf16 values + one RNE rounding = the correctly rounded bf16 add (24 >= 2*8+2 bits)
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// XCD-aware block remap: hardware places block b on XCD b % 8; give each XCD a contiguous
// range of logical tiles so the tiles of one (layer, head) - which share K - share an L2.
__device__ __forceinline__ int logical_block(int total_wgs, int wgs_per_xcd) {
    const int b = blockIdx.x;
    const int l = (b & 7) * wgs_per_xcd + (b >> 3);
    return l < total_wgs ? l : -1;
}

__device__ __forceinline__ int find_layer(const TapLayer* layers, int n, int wg) {
    int lo = 0, hi = n - 1;                       // last layer with wg_begin <= wg
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (layers[mid].wg_begin <= wg) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------
// Generic tap.  256 threads = 64 pixels x 4 token chunks.  Thread (p, part) owns the running
// sums acc[t][p0 + p] for its <= 20 tokens in registers across every step of the launch.
// LDS: K of the current (step, head) as f32 [tokens][d] (broadcast reads), plus the
// cross-chunk softmax reductions.
// ---------------------------------------------------------------------------------------
template <typename IN_T, typename ACC_T>
__global__ __launch_bounds__(256) void tap_generic_kernel(const TapLaunch L)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int wg = logical_block(L.total_wgs, L.wgs_per_xcd);
    if (wg < 0) return;

    TapLayer lay;
    const TapPtr* ptrs;
    if (L.layers) {
        lay = L.layers[find_layer(L.layers, L.n_layers, wg)];
        ptrs = L.ptrs + lay.ptr_begin;
    } else {
        lay = L.one;
        ptrs = &L.one_ptr;
    }
    const int tokens = L.tokens;
    const int d = lay.head_dim;
    const int rel = wg - lay.wg_begin;
    const int kh = rel / lay.tiles_per_head;              // kept-head index 0..heads_kept-1
    const int p0 = (rel - kh * lay.tiles_per_head) * kTapPixels;
    const int bh = lay.bh_first + kh;
    const int b = bh / lay.heads, h = bh - b * lay.heads;

    float* Ks = reinterpret_cast<float*>(smem_raw);        // [tokens][d]
    float* red = Ks + tokens * d;                          // [2][kTapParts][kTapPixels]

    const int tid = threadIdx.x;
    const int p = tid & (kTapPixels - 1);
    const int part = tid >> 6;
    const int pixel = p0 + p;
    const bool valid = pixel < lay.hw;
    const int t0 = part * kTokPerPart;
    const int nt = min(kTokPerPart, tokens - t0);          // may be <= 0 for tiny token counts

    ACC_T* acc = reinterpret_cast<ACC_T*>(lay.acc) + (size_t)kh * tokens * lay.hw;
    float run[kTokPerPart];
#pragma unroll
    for (int i = 0; i < kTokPerPart; ++i)
        run[i] = (valid && i < nt) ? ld<ACC_T>(acc + (size_t)(t0 + i) * lay.hw + pixel) : 0.f;

    for (int s = 0; s < lay.n_steps; ++s) {
        const IN_T* q = reinterpret_cast<const IN_T*>(ptrs[s].q) + b * lay.q_sb + h * lay.q_sh;
        const IN_T* k = reinterpret_cast<const IN_T*>(ptrs[s].k) + b * lay.k_sb + h * lay.k_sh;
        __syncthreads();                                   // previous step done with Ks / red
        for (int i = tid; i < tokens * d; i += 256) {
            const int t = i / d, dd = i - t * d;
            Ks[i] = ld<IN_T>(k + t * lay.k_st + dd);
        }
        __syncthreads();

        float logit[kTokPerPart];
#pragma unroll
        for (int i = 0; i < kTokPerPart; ++i) logit[i] = 0.f;
        if (valid) {
            const IN_T* qrow = q + (int64_t)pixel * lay.q_sp;
            for (int dd = 0; dd < d; ++dd) {
                const float qv = ld<IN_T>(qrow + dd);
#pragma unroll
                for (int i = 0; i < kTokPerPart; ++i)
                    if (i < nt) logit[i] = fmaf(qv, Ks[(t0 + i) * d + dd], logit[i]);
            }
        }
        // alpha applied in f32 to the f32 dot product, then the baddbmm output rounding
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < kTokPerPart; ++i) {
            float x = logit[i] * lay.scale;
            if (lay.round_logits) x = round_to<IN_T>(x);
            logit[i] = x;
            if (i < nt) m = fmaxf(m, x);
        }
        red[part * kTapPixels + p] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[p], red[kTapPixels + p]), fmaxf(red[2 * kTapPixels + p], red[3 * kTapPixels + p]));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < kTokPerPart; ++i) {
            const float e = (i < nt) ? expf(logit[i] - m) : 0.f;
            logit[i] = e;
            sum += e;
        }
        float* rsum = red + kTapParts * kTapPixels;
        rsum[part * kTapPixels + p] = sum;
        __syncthreads();
        sum = (rsum[p] + rsum[kTapPixels + p]) + (rsum[2 * kTapPixels + p] + rsum[3 * kTapPixels + p]);
#pragma unroll
        for (int i = 0; i < kTokPerPart; ++i) {
            const float prob = round_to<IN_T>(logit[i] / sum);       // probs.to(dtype)
            run[i] = round_to<ACC_T>(run[i] + prob);                  // heatmap.py:156
        }
    }
    if (valid) {
#pragma unroll
        for (int i = 0; i < kTokPerPart; ++i)
            if (i < nt) st<ACC_T>(acc + (size_t)(t0 + i) * lay.hw + pixel, run[i]);
    }
}

// ---------------------------------------------------------------------------------------
// Tap from materialised probabilities [BH, hw, tokens] (contiguous).  The 64-pixel chunk of
// one head is one contiguous run of 64*tokens elements: stage it through LDS with coalesced
// reads, then the same (pixel, token-chunk) ownership as above does the transposed add.
// ---------------------------------------------------------------------------------------
template <typename IN_T, typename ACC_T>
__global__ __launch_bounds__(256) void tap_probs_kernel(const ProbsLaunch L)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int wg = logical_block(L.total_wgs, L.wgs_per_xcd);
    if (wg < 0) return;
    const int tokens = L.tokens;
    const int kh = wg / L.tiles_per_head;
    const int p0 = (wg - kh * L.tiles_per_head) * kTapPixels;
    const int npix = min(kTapPixels, L.hw - p0);
    IN_T* raw = reinterpret_cast<IN_T*>(smem_raw);         // [npix][tokens]
    const IN_T* src = reinterpret_cast<const IN_T*>(L.probs) + ((size_t)(L.bh_first + kh) * L.hw + p0) * tokens;
    const int tid = threadIdx.x;
    const int n = npix * tokens;
    for (int i = tid; i < n; i += 256) raw[i] = src[i];
    __syncthreads();
    const int p = tid & (kTapPixels - 1);
    const int part = tid >> 6;
    if (p >= npix) return;
    const int t0 = part * kTokPerPart;
    const int nt = min(kTokPerPart, tokens - t0);
    ACC_T* acc = reinterpret_cast<ACC_T*>(L.acc) + (size_t)kh * tokens * L.hw + (p0 + p);
    for (int i = 0; i < nt; ++i) {
        ACC_T* a = acc + (size_t)(t0 + i) * L.hw;
        st<ACC_T>(a, ld<ACC_T>(a) + ld<IN_T>(raw + p * tokens + t0 + i));
    }
}

// ---------------------------------------------------------------------------------------
// Finalize: grid (tokens, n_chunks).  Workgroup (t, c) walks keys c, c + n_chunks, ... :
// plane -> LDS (f32) -> x pass -> y pass -> clamp -> += LDS out tile; one f32 atomicAdd per
// output element per workgroup at the end (scaled by 1 / n_keys).
// Separable in the same order as torch's upsample_bicubic2d (x on the 4 source rows, then y).
// ---------------------------------------------------------------------------------------
template <typename ACC_T>
__global__ __launch_bounds__(256) void finalize_kernel(const FinLaunch L)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tok = blockIdx.x;
    const int chunk = blockIdx.y;
    const int O = L.out_side;
    float* outt = reinterpret_cast<float*>(smem_raw);            // [O][O]
    float* plane = outt + O * O;                                 // [max_side][max_side]
    float* tmp = plane + L.max_side * L.max_side;                // [max_side][O]
    const int tid = threadIdx.x;
    for (int i = tid; i < O * O; i += 256) outt[i] = 0.f;

    for (int kidx = chunk; kidx < L.n_keys; kidx += L.n_chunks) {
        const FinKey key = L.keys[kidx];
        const int S = key.side;
        const ACC_T* src = reinterpret_cast<const ACC_T*>(key.base) + (size_t)tok * S * S;
        if (key.tab < 0) {                                       // same size: copy (+ clamp)
            for (int i = tid; i < O * O; i += 256) outt[i] += fmaxf(ld<ACC_T>(src + i), 0.f);
            continue;
        }
        const int16_t* tix = L.tab_idx + (size_t)key.tab * O * 4;
        const float* tw = L.tab_w + (size_t)key.tab * O * 4;
        __syncthreads();                                         // previous key done with plane/tmp
        for (int i = tid; i < S * S; i += 256) plane[i] = ld<ACC_T>(src + i);
        __syncthreads();
        for (int i = tid; i < S * O; i += 256) {
            const int y = i / O, ox = i - y * O;
            const float* row = plane + y * S;
            const int16_t* ix = tix + ox * 4;
            const float* w = tw + ox * 4;
            tmp[i] = row[ix[0]] * w[0] + row[ix[1]] * w[1] + row[ix[2]] * w[2] + row[ix[3]] * w[3];
        }
        __syncthreads();
        for (int i = tid; i < O * O; i += 256) {
            const int oy = i / O, ox = i - oy * O;
            const int16_t* iy = tix + oy * 4;
            const float* w = tw + oy * 4;
            const float v = tmp[iy[0] * O + ox] * w[0] + tmp[iy[1] * O + ox] * w[1] +
                            tmp[iy[2] * O + ox] * w[2] + tmp[iy[3] * O + ox] * w[3];
            outt[i] += fmaxf(v, 0.f);
        }
    }
    // each thread only ever touched its own outt[i] entries (i = tid mod 256): no barrier needed
    float* out = L.out + (size_t)tok * O * O;
    for (int i = tid; i < O * O; i += 256) atomicAdd(out + i, outt[i] * L.inv_n);
}

// trace.py:129-130  maps[:n] / (maps[1:n-1].sum(0) + 1e-6)
__global__ __launch_bounds__(256) void normalize_kernel(float* maps, int n_rows, int plane)
{
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= plane) return;
    float s = 0.f;
    for (int t = 1; t < n_rows - 1; ++t) s += maps[(size_t)t * plane + px];
    s += 1e-6f;
    for (int t = 0; t < n_rows; ++t) maps[(size_t)t * plane + px] /= s;
}

// ---------------------------------------------------------------------------------------
// Word heat map (heatmap.py:121-123) and expand_as (heatmap.py:77-93).
// ---------------------------------------------------------------------------------------
struct WordIdx { int32_t n; int32_t idx[kMaxTokens]; };

__global__ __launch_bounds__(256) void word_mean_kernel(const float* maps, int plane, WordIdx w, float* word_map,
                                                        float* minmax)
{
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // order-preserving int encodings of +inf / -inf for the atomicMin / atomicMax below
        reinterpret_cast<int*>(minmax)[0] = 0x7f800000;
        reinterpret_cast<int*>(minmax)[1] = (int)0x80000000 ^ 0x7fffffff ^ 0x7f800000;   // enc(-inf)
    }
    if (px >= plane) return;
    float s = 0.f;
    for (int i = 0; i < w.n; ++i) s += maps[(size_t)w.idx[i] * plane + px];
    word_map[px] = s / (float)w.n;
}

__device__ __forceinline__ int enc_ordered(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float dec_ordered(int i) {
    return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff);
}

__device__ __forceinline__ void cubic_coeffs(float t, float w[4]) {
#pragma clang fp contract(off)
    const float A = -0.75f;
    const float x0 = t + 1.0f;
    w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
    const float u = 1.0f - t;
    w[2] = ((A + 2.0f) * u - (A + 3.0f)) * u * u + 1.0f;
    const float x3 = u + 1.0f;
    w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

// Results the caller copies to the host right behind the kernel (expand_as returns a CPU tensor, heatmap.py:88) are stored write-through
// (system scope): the copy engine reads memory, not the L2s.  Round 6 saw ONE expand_as result in ~10^5 whose 64 consecutive elements (two
// cache lines) still held the block's previous owner's data after the copy (four test processes sharing the GPU; not reproduced in 80 000
// calls) -- with write-through stores the result does not depend on when an L2 writes a dirty line back.
__device__ __forceinline__ void store_for_host(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// one thread per output pixel; source plane (<= 96x96 f32) is L1/L2 resident
__global__ __launch_bounds__(256) void word_expand_kernel(const float* word_map, int side, float* out, int out_h,
                                                          int out_w, float* minmax)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    const bool valid = i < out_h * out_w;
    if (valid) {
        const int oy = i / out_w, ox = i - oy * out_w;
        float wy[4], wx[4];
        int iy[4], ix[4];
        {
            const float sc = (float)side / (float)out_h;
            const float src = sc * ((float)oy + 0.5f) - 0.5f;
            const float f = floorf(src);
            cubic_coeffs(src - f, wy);
            for (int a = 0; a < 4; ++a) iy[a] = min(max((int)f - 1 + a, 0), side - 1);
        }
        {
            const float sc = (float)side / (float)out_w;
            const float src = sc * ((float)ox + 0.5f) - 0.5f;
            const float f = floorf(src);
            cubic_coeffs(src - f, wx);
            for (int a = 0; a < 4; ++a) ix[a] = min(max((int)f - 1 + a, 0), side - 1);
        }
        if (side == out_h && side == out_w) {
            v = word_map[i];
        } else {
            float rows[4];
            for (int a = 0; a < 4; ++a) {
                const float* r = word_map + iy[a] * side;
                rows[a] = r[ix[0]] * wx[0] + r[ix[1]] * wx[1] + r[ix[2]] * wx[2] + r[ix[3]] * wx[3];
            }
            v = rows[0] * wy[0] + rows[1] * wy[1] + rows[2] * wy[2] + rows[3] * wy[3];
        }
        store_for_host(out + i, v);
    }
    // wave64 min / max, one atomic pair per wave
    float lo = valid ? v : INFINITY, hi = valid ? v : -INFINITY;
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off, 64));
        hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(reinterpret_cast<int*>(minmax), enc_ordered(lo));
        atomicMax(reinterpret_cast<int*>(minmax) + 1, enc_ordered(hi));
    }
}

__global__ __launch_bounds__(256) void word_post_kernel(float* out, int n, const float* minmax, int absolute,
                                                        float threshold)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = out[i];
    if (!absolute) {
        const float lo = dec_ordered(reinterpret_cast<const int*>(minmax)[0]);
        const float hi = dec_ordered(reinterpret_cast<const int*>(minmax)[1]);
        v = (v - lo) / (hi - lo + 1e-8f);
    }
    if (threshold != 0.f) v = v > threshold ? 1.f : 0.f;      // `if threshold:` (heatmap.py:85)
    store_for_host(out + i, v);
}

// ---------------------------------------------------------------------------------------
// evaluate.compute_iou / compute_ioa (reference daam/evaluate.py:14-35) for a batch of (prediction, truth) pairs.
// One thread per pixel of the truth mask b; when the shapes differ (the reference tests shape[0] only) the prediction a is
// resized with the bicubic of F.interpolate (align_corners=False, A = -0.75, border-clamped taps; x on the four source rows,
// then y) and binarised (a < 1 -> 0, else 1); sums[pair] += {a*b, a, b}, one f32 atomic per wave and quantity -- exact for
// binary masks (integer sums below 2^24), order-dependent in the last bits for soft ones.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_overlap_kernel(const float* a, int a_h, int a_w, const float* b, int b_h, int b_w,
                                                           int resize, float* sums)
{
#pragma clang fp contract(off)
    const int pair = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float* ap = a + (size_t)pair * a_h * a_w;
    float va = 0.f, vb = 0.f;
    if (i < b_h * b_w) {
        vb = b[(size_t)pair * b_h * b_w + i];
        if (!resize) {
            va = ap[i];
        } else {
            const int oy = i / b_w, ox = i - oy * b_w;
            float wy[4], wx[4];
            int iy[4], ix[4];
            {
                const float sc = (float)a_h / (float)b_h;
                const float src = sc * ((float)oy + 0.5f) - 0.5f;
                const float f = floorf(src);
                cubic_coeffs(src - f, wy);
                for (int t = 0; t < 4; ++t) iy[t] = min(max((int)f - 1 + t, 0), a_h - 1);
            }
            {
                const float sc = (float)a_w / (float)b_w;
                const float src = sc * ((float)ox + 0.5f) - 0.5f;
                const float f = floorf(src);
                cubic_coeffs(src - f, wx);
                for (int t = 0; t < 4; ++t) ix[t] = min(max((int)f - 1 + t, 0), a_w - 1);
            }
            float rows[4];
            for (int t = 0; t < 4; ++t) {
                const float* r = ap + (size_t)iy[t] * a_w;
                rows[t] = r[ix[0]] * wx[0] + r[ix[1]] * wx[1] + r[ix[2]] * wx[2] + r[ix[3]] * wx[3];
            }
            const float v = rows[0] * wy[0] + rows[1] * wy[1] + rows[2] * wy[2] + rows[3] * wy[3];
            va = v < 1.0f ? 0.0f : (v >= 1.0f ? 1.0f : v);    // a[a < 1] = 0; a[a >= 1] = 1  (evaluate.py:17-18): a NaN stays a NaN
        }
    }
    float inter = va * vb, sa = va, sb = vb;
    for (int off = 32; off > 0; off >>= 1) {
        inter += __shfl_xor(inter, off, 64);
        sa += __shfl_xor(sa, off, 64);
        sb += __shfl_xor(sb, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(sums + 3 * pair + 0, inter);
        atomicAdd(sums + 3 * pair + 1, sa);
        atomicAdd(sums + 3 * pair + 2, sb);
    }
}

hipError_t launch_mask_overlap(const float* a, int a_h, int a_w, const float* b, int b_h, int b_w, int n, float* sums, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(float) * 3 * (size_t)n, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mask_overlap_kernel, dim3((b_h * b_w + 255) / 256, n), dim3(256), 0, stream, a, a_h, a_w, b, b_h, b_w,
                       a_h != b_h ? 1 : 0, sums);
    return hipGetLastError();
}

// per-launch device tables: pinned host (device-mapped) -> device twin, in stream order on the
// compute queue (16 bytes per thread; tables are a few tens of KB)
__global__ __launch_bounds__(256) void upload_kernel(float4* dst, const float4* src, int n16, float4* zero, int z16)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
    else if (i - n16 < z16) zero[i - n16] = float4{0.f, 0.f, 0.f, 0.f};
}

// `zero` / `zero_bytes` (16-byte multiple, may be 0): a buffer cleared by the same launch (the finalize
// output, which is accumulated with atomics) -- saves a separate memset on the critical path
hipError_t launch_upload(void* dst, const void* src_host_mapped, size_t bytes, void* zero, size_t zero_bytes,
                         hipStream_t stream)
{
    const int n16 = (int)((bytes + 15) / 16);                 // ring slots are 256-byte aligned and padded
    const int z16 = (int)(zero_bytes / 16);
    hipLaunchKernelGGL(upload_kernel, dim3((n16 + z16 + 255) / 256), dim3(256), 0, stream, reinterpret_cast<float4*>(dst),
                       reinterpret_cast<const float4*>(src_host_mapped), n16, reinterpret_cast<float4*>(zero), z16);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// host-callable launchers (called from daam_api.hip)
// ---------------------------------------------------------------------------------------
template <typename K> static hipError_t allow_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

hipError_t launch_tap_generic(const TapLaunch& L, int in_dtype, int acc_dtype, int max_d, hipStream_t stream,
                              int* grid_out, int* lds_out)
{
    const size_t lds = (size_t)L.tokens * max_d * sizeof(float) + 2 * kTapParts * kTapPixels * sizeof(float);
    const int grid = L.wgs_per_xcd * 8;
    *grid_out = grid;
    *lds_out = (int)lds;
    hipError_t e;
#define DAAM_LAUNCH(IN, ACC)                                                            \
    do {                                                                                \
        if ((e = allow_lds(tap_generic_kernel<IN, ACC>, lds)) != hipSuccess) return e;  \
        hipLaunchKernelGGL((tap_generic_kernel<IN, ACC>), dim3(grid), dim3(256), lds, stream, L); \
    } while (0)
    if (in_dtype == 0 && acc_dtype == 0) DAAM_LAUNCH(__half, __half);
    else if (in_dtype == 0 && acc_dtype == 1) DAAM_LAUNCH(__half, float);
    else if (in_dtype == 2 && acc_dtype == 2) DAAM_LAUNCH(bf16_t, bf16_t);
    else if (in_dtype == 2 && acc_dtype == 1) DAAM_LAUNCH(bf16_t, float);
    else if (in_dtype == 1 && acc_dtype == 1) DAAM_LAUNCH(float, float);
    else return hipErrorInvalidValue;
#undef DAAM_LAUNCH
    return hipGetLastError();
}

hipError_t launch_tap_probs(const ProbsLaunch& L, int in_dtype, int acc_dtype, hipStream_t stream, int* grid_out,
                            int* lds_out)
{
    const size_t lds = (size_t)kTapPixels * L.tokens * (in_dtype == 1 ? 4 : 2);
    const int grid = L.wgs_per_xcd * 8;
    *grid_out = grid;
    *lds_out = (int)lds;
    if (in_dtype == 0 && acc_dtype == 0)
        hipLaunchKernelGGL((tap_probs_kernel<__half, __half>), dim3(grid), dim3(256), lds, stream, L);
    else if (in_dtype == 0 && acc_dtype == 1)
        hipLaunchKernelGGL((tap_probs_kernel<__half, float>), dim3(grid), dim3(256), lds, stream, L);
    else if (in_dtype == 2 && acc_dtype == 2)
        hipLaunchKernelGGL((tap_probs_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), lds, stream, L);
    else if (in_dtype == 2 && acc_dtype == 1)
        hipLaunchKernelGGL((tap_probs_kernel<bf16_t, float>), dim3(grid), dim3(256), lds, stream, L);
    else if (in_dtype == 1 && acc_dtype == 1)
        hipLaunchKernelGGL((tap_probs_kernel<float, float>), dim3(grid), dim3(256), lds, stream, L);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_finalize(const FinLaunch& L, int acc_dtype, hipStream_t stream, int* grid_out, int* lds_out)
{
    const size_t lds = sizeof(float) * ((size_t)L.out_side * L.out_side + (size_t)L.max_side * L.max_side +
                                        (size_t)L.max_side * L.out_side);
    *grid_out = L.tokens * L.n_chunks;
    *lds_out = (int)lds;
    hipError_t e;
    if (acc_dtype == 0) {
        if ((e = allow_lds(finalize_kernel<__half>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((finalize_kernel<__half>), dim3(L.tokens, L.n_chunks), dim3(256), lds, stream, L);
    } else if (acc_dtype == 2) {
        if ((e = allow_lds(finalize_kernel<bf16_t>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((finalize_kernel<bf16_t>), dim3(L.tokens, L.n_chunks), dim3(256), lds, stream, L);
    } else {
        if ((e = allow_lds(finalize_kernel<float>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((finalize_kernel<float>), dim3(L.tokens, L.n_chunks), dim3(256), lds, stream, L);
    }
    return hipGetLastError();
}

// Shader-clock monitor (daam_clock_monitor_start): ONE wave; lane 0 stores (shader cycles, 100 MHz reference ticks) every
// `period_us`, sleeping in between (s_sleep keeps it off the issue ports of the kernels under test).
__global__ __launch_bounds__(64) void clock_monitor_kernel(unsigned long long* samples, int n_samples, int period_us)
{
    if (threadIdx.x != 0) return;
    const unsigned long long period = (unsigned long long)period_us * 100ull;       // reference ticks
    unsigned long long next = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n_samples; ++i) {
        while (__builtin_amdgcn_s_memrealtime() < next) __builtin_amdgcn_s_sleep(32);
        const unsigned long long cyc = __builtin_amdgcn_s_memtime();
        const unsigned long long ref = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_store(samples + 2 * i, cyc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(samples + 2 * i + 1, ref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        next = ref + period;
    }
}

hipError_t launch_clock_monitor(unsigned long long* samples, int n_samples, int period_us, hipStream_t stream)
{
    hipLaunchKernelGGL(clock_monitor_kernel, dim3(1), dim3(64), 0, stream, samples, n_samples, period_us);
    return hipGetLastError();
}

// Start gate of a multi-kernel tap flush (SD-v1.5: head_dim 40 / 80 / 160 = three kernels).  The small kernels run on side
// streams behind an event wait, the large one on the caller's stream -- and when the large one's 1024 workgroups are dispatched
// first they take every CU's LDS for their whole 50-step life and the small kernels only start when they drain (measured: 495 us
// per flush against 435 us when the small kernels' workgroups are resident first and the large grid fills in around them).  One
// wave on the caller's stream, ahead of the large kernel: wait until the side kernels' workgroups have counted themselves in
// (TapLaunch::started), or for `timeout_us` -- it holds no LDS and one wave slot, so it can never keep them from starting.
// A gate that runs into its timeout (the auxiliary streams did NOT run beside the caller's: e.g. mapped onto one hardware queue)
// says so in `timeouts` (pinned host memory, device-mapped): the host then stops using the gate for the context.
__global__ __launch_bounds__(64) void start_gate_kernel(const unsigned* counter, unsigned target, int timeout_us, unsigned* timeouts)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                 // 100 MHz
    const unsigned long long limit = (unsigned long long)timeout_us * 100ull;
    while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > limit) {
            if (timeouts && threadIdx.x == 0) __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(4);
    }
}

hipError_t launch_start_gate(const unsigned* counter, unsigned target, int timeout_us, unsigned* timeouts, hipStream_t stream)
{
    hipLaunchKernelGGL(start_gate_kernel, dim3(1), dim3(64), 0, stream, counter, target, timeout_us, timeouts);
    return hipGetLastError();
}

hipError_t launch_normalize(float* maps, int n_rows, int plane, hipStream_t stream)
{
    hipLaunchKernelGGL(normalize_kernel, dim3((plane + 255) / 256), dim3(256), 0, stream, maps, n_rows, plane);
    return hipGetLastError();
}

hipError_t launch_word(const float* maps, int side, const int32_t* idx, int n_idx, float* word_map, float* out,
                       int out_h, int out_w, int absolute, float threshold, float* workspace, hipStream_t stream)
{
    WordIdx w;
    w.n = n_idx;
    for (int i = 0; i < n_idx; ++i) w.idx[i] = idx[i];
    const int plane = side * side;
    hipLaunchKernelGGL(word_mean_kernel, dim3((plane + 255) / 256), dim3(256), 0, stream, maps, plane, w, word_map,
                       workspace);
    if (out) {
        const int n = out_h * out_w;
        hipLaunchKernelGGL(word_expand_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, word_map, side, out,
                           out_h, out_w, workspace);
        if (!absolute || threshold != 0.f)
            hipLaunchKernelGGL(word_post_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, out, n, workspace,
                               absolute, threshold);
    }
    return hipGetLastError();
}

}  // namespace daam
