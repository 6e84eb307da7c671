// Shared pieces of the two MFMA tap kernels (daam_tap_mfma.hip: 32x32x16 tiles, any head_dim multiple of 8;
// daam_tap_d64.hip: 16x16x32 tiles, head_dim <= 64).
#pragma once
#include "daam_types.h"

namespace daam {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef unsigned short ushort8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int kTok = 77;               // context_size (trace.py:194); the only value the reference taps
constexpr int kTokRows = 96;           // 3 MFMA row tiles
constexpr int kMfmaPixels = 128;       // pixels per workgroup
constexpr int kSlots = 40;             // token slots per lane: 16 + 16 + 8

__device__ __forceinline__ int mfma_logical_block(int total_wgs, int wgs_per_xcd) {
    const int b = blockIdx.x;
    const int l = (b & 7) * wgs_per_xcd + (b >> 3);
    return l < total_wgs ? l : -1;
}

// start gate of a multi-kernel flush: a side kernel's workgroups count themselves in (see start_gate_kernel, daam_kernels.hip)
__device__ __forceinline__ void tap_mark_started(const TapLaunch& L) {
    if (L.started && threadIdx.x == 0) __hip_atomic_fetch_add(L.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int mfma_find_layer(const DAAM_GLOBAL TapLayer* layers, int n, int wg) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (layers[mid].wg_begin <= wg) lo = mid; else hi = mid - 1;
    }
    return lo;
}

constexpr int kMaxStepsPerLaunch = 64;    // per-layer step pointers staged in LDS (1 KiB)

__device__ __forceinline__ void load_layer(const DAAM_GLOBAL TapLayer* g, TapLayer* out) {
    out->acc = g->acc; out->heads_kept = g->heads_kept; out->bh_first = g->bh_first; out->heads = g->heads;
    out->hw = g->hw; out->head_dim = g->head_dim; out->tiles_per_head = g->tiles_per_head;
    out->wg_begin = g->wg_begin; out->n_steps = g->n_steps; out->ptr_begin = g->ptr_begin;
    out->round_logits = g->round_logits; out->scale = g->scale; out->fresh = g->fresh;
    out->q_sb = g->q_sb; out->q_sh = g->q_sh; out->q_sp = g->q_sp;
    out->k_sb = g->k_sb; out->k_sh = g->k_sh; out->k_st = g->k_st;
    out->px_begin = g->px_begin; out->px_end = g->px_end; out->tile_px = g->tile_px;
}

constexpr float kMasked = -1.0e30f;    // logit of the padding tokens: exp() underflows to exactly 0

// e^d for d <= 0 to ~1 ulp: 2^(d*log2e) on v_exp_f32 with the rounding error of the product
// (and the low part of log2e) folded back in:  2^t * (1 + err * ln2).
__device__ __forceinline__ float exp_nonpos(float d) {
    const float L = 1.44269502162933349609375f;       // log2(e) rounded to f32
    const float Llo = 1.92596303e-08f;                // log2(e) - L
    const float t = d * L;
    float err = __builtin_fmaf(d, L, -t);
    err = __builtin_fmaf(d, Llo, err);
    const float r = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(r, err * 0.693147182f, r);
}

// f32 pair -> packed fp16, round-to-nearest-even, as ONE v_cvt_pk_f16_f32.  Left to itself the
// compiler folds the preceding multiply into v_fma_mixlo_f16 / v_fma_mixhi_f16, which measured
// 3.4 ns per wave-instruction per SIMD against 1.8 ns for v_cvt_pk_f16_f32 (tools/ubench_valu.hip)
// and handles one element instead of two.
__device__ __forceinline__ half2v cvt_pk_rne(float2v v) {
    half2v r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(v[0]), "v"(v[1]));
    return r;
}

// packed fp16 max without the canonicalising v_pk_max_f16 x, x the compiler puts in front of
// __builtin_elementwise_max when it cannot prove its inputs are not signalling NaNs
__device__ __forceinline__ half2v pk_max(half2v a, half2v b) {
    half2v r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// token of slot i for lane half g:  C/D row = (reg&3) + 8*(reg>>2) + 4*g  (+32 per row tile)
__device__ __forceinline__ constexpr int slot_token(int i, int g) {
    const int mt = i >> 4, reg = i & 15;
    return mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * g;
}

template <typename ACC_T> struct AccVec;
template <> struct AccVec<_Float16> { static constexpr int kPerVec = 8; };
template <> struct AccVec<float> { static constexpr int kPerVec = 4; };


// ---------------------------------------------------------------------------------------
// softmax over the 77 tokens of this lane's pixel + accumulate into the running sums.
// c0 / c1 / c2 = the three 32-token MFMA row tiles (f32 q.k), g = lane half, run = 40 slots.
// Rounding points of the reference: fp16( f32(q.k) * scale ) -> f32 softmax -> fp16(p) -> add.
// ---------------------------------------------------------------------------------------
template <typename ACC_T, bool FAST_EXP>
__device__ __forceinline__ void softmax_accumulate(const floatx16& c0, const floatx16& c1, const floatx16& c2,
                                                   const TapLayer& lay, int g, ACC_T (&run)[kSlots])
{
        if constexpr (FAST_EXP) {
            // Fast softmax (host guarantees round_logits): logits stay packed fp16 (that IS their
            // reference precision), max on v_pk_max_f16, exponent argument by ONE mixed-precision FMA
            // t = x*log2(e) - m*log2(e) straight from the fp16 logit (v_fma_mix_f32), 2^t on v_exp_f32.
            // The rounding of m*log2(e) is common to the 77 tokens of a pixel and cancels in e/sum; what
            // remains (one f32 rounding of t, log2(e) to f32) is <~1e-6 relative, i.e. the same class of
            // deviation as the f32 summation order of q.k (an occasional 1-ulp flip of an fp16 probability).
            half2v xh[kSlots / 2];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xh[i] = cvt_pk_rne(float2v{c0[2 * i], c0[2 * i + 1]} * lay.scale);
                xh[8 + i] = cvt_pk_rne(float2v{c1[2 * i], c1[2 * i + 1]} * lay.scale);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xh[16 + i] = cvt_pk_rne(float2v{c2[2 * i], c2[2 * i + 1]} * lay.scale);
            if (g == 1) {                                                // tokens 77..79 of the upper lane half
                const _Float16 ninf = -(_Float16)__builtin_inff();
                xh[18][1] = ninf;
                xh[19] = half2v{ninf, ninf};
            }
            const float L = 1.44269502162933349609375f;
            float2v ev[kSlots / 2];
            auto exps = [&](float nmL) -> float {                        // e = 2^(x L - m L), returns the pixel's sum
                float2v sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < kSlots / 2; i += 2) {
                    ev[i] = float2v{__builtin_amdgcn_exp2f(__builtin_fmaf((float)xh[i][0], L, nmL)),
                                    __builtin_amdgcn_exp2f(__builtin_fmaf((float)xh[i][1], L, nmL))};
                    ev[i + 1] = float2v{__builtin_amdgcn_exp2f(__builtin_fmaf((float)xh[i + 1][0], L, nmL)),
                                        __builtin_amdgcn_exp2f(__builtin_fmaf((float)xh[i + 1][1], L, nmL))};
                    sa += ev[i];
                    sb += ev[i + 1];
                }
                sa += sb;
                const float sum = sa[0] + sa[1];
                return sum + __shfl_xor(sum, 32, 64);
            };
            // Reference point = token 0's logit (lane half 0, slot 0; its own term is exactly 1, so the sum cannot
            // underflow) instead of the row maximum -- softmax is shift-invariant.  Only when some logit exceeds it
            // by more than ~69 (sum > 2^100: 1/sum would leave the normal f32 range; past 88 the exponentials
            // overflow) is the row redone with the true maximum, as the reference does.
            const auto x0 = __builtin_amdgcn_permlane32_swap(__float_as_uint((float)xh[0][0]), __float_as_uint((float)xh[0][0]),
                                                             false, false);      // x0[0] = lanes (lo, lo)
            float tot = exps(-__uint_as_float(x0[0]) * L);
            if (__builtin_expect(!(tot <= 0x1p100f), 0)) {               // large, inf or NaN
                half2v ma = xh[0], mb = xh[1];
#pragma unroll
                for (int i = 2; i < kSlots / 2; i += 2) {
                    ma = pk_max(ma, xh[i]);
                    mb = pk_max(mb, xh[i + 1]);
                }
                ma = pk_max(ma, mb);
                float m = fmaxf((float)ma[0], (float)ma[1]);
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                tot = exps(-m * L);
            }
            const float inv = __builtin_amdgcn_rcpf(tot);                  // v_rcp_f32: 1 ulp
#pragma unroll
            for (int i = 0; i < kSlots / 2; ++i) {
                const half2v ph = cvt_pk_rne(ev[i] * inv);                // probs.to(dtype)
                if constexpr (sizeof(ACC_T) == 2) {                      // heatmap.py:156, as v_pk_add_f16
                    half2v r = {(_Float16)run[2 * i], (_Float16)run[2 * i + 1]};
                    r += ph;
                    run[2 * i] = (ACC_T)r[0];
                    run[2 * i + 1] = (ACC_T)r[1];
                } else {
                    run[2 * i] = run[2 * i] + (ACC_T)ph[0];
                    run[2 * i + 1] = run[2 * i + 1] + (ACC_T)ph[1];
                }
            }
        } else {
        // logits: alpha in f32, then the baddbmm output rounding (skipped for upcast_attention)
        float x[kSlots];
#pragma unroll
        for (int i = 0; i < 16; ++i) { x[i] = c0[i] * lay.scale; x[16 + i] = c1[i] * lay.scale; }
#pragma unroll
        for (int i = 0; i < 8; ++i) x[32 + i] = c2[i] * lay.scale;
        if (lay.round_logits) {
#pragma unroll
            for (int i = 0; i < kSlots; ++i) x[i] = (float)(_Float16)x[i];
        }
        if (g == 1) { x[37] = kMasked; x[38] = kMasked; x[39] = kMasked; }   // tokens 77..79 of the upper lane half
        float m0 = x[0], m1 = x[1], m2 = x[2], m3 = x[3];
#pragma unroll
        for (int i = 4; i < kSlots; i += 4) {
            m0 = fmaxf(m0, x[i]); m1 = fmaxf(m1, x[i + 1]); m2 = fmaxf(m2, x[i + 2]); m3 = fmaxf(m3, x[i + 3]);
        }
        float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < kSlots; i += 4) {
            x[i] = exp_nonpos(x[i] - m);         s0 += x[i];
            x[i + 1] = exp_nonpos(x[i + 1] - m); s1 += x[i + 1];
            x[i + 2] = exp_nonpos(x[i + 2] - m); s2 += x[i + 2];
            x[i + 3] = exp_nonpos(x[i + 3] - m); s3 += x[i + 3];
        }
        float sum = (s0 + s1) + (s2 + s3);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < kSlots; ++i) {
            const _Float16 prob = (_Float16)(x[i] * inv);                // probs.to(dtype)
            run[i] = run[i] + (ACC_T)prob;                               // heatmap.py:156
        }
        }
}

}  // namespace daam
