// Softmax + running-sum update of the 16x16x32 tiling (20 token slots per lane), shared by daam_tap_d64.hip (head_dim <= 64)
// and daam_tap_wide.hip (head_dim <= 160).
#pragma once
#include "daam_tap16.h"

namespace daam {

// softmax over the 77 tokens of the lane's pixel (20 slots here, 57 in the three partner lanes)
// + accumulate.  c[mt][r] = f32 q.k of token 16mt + 4h + r.
// a POSITIVE, normal power of two (attention scales are: head_dim^-1/2 of 64, 16, 256): only then may the scale be folded into the exponent
// constant / the logits stay unscaled in the overflow redo (a negative, zero, inf or NaN "scale" takes the general branch, which handles any sign)
__device__ __forceinline__ bool scale_is_pow2(float scale) {
    const unsigned b = __float_as_uint(scale);
    return (b & 0x807fffffu) == 0 && b != 0 && b < 0x7f800000u;
}

template <typename ACC_T> struct Pair;
template <> struct Pair<_Float16> { using T = half2v; };
template <> struct Pair<float> { using T = float2v; };
template <> struct Pair<bf16_t> { using T = float2v; };      // bf16 sums live in registers as f32 holding bf16 values
template <> struct AccVec<bf16_t> { static constexpr int kPerVec = 8; };

// running-sum element <-> its register representation
template <typename ACC_T> __device__ __forceinline__ auto from_acc(ACC_T v) { return v; }
template <> __device__ __forceinline__ auto from_acc<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <typename ACC_T, typename R> __device__ __forceinline__ ACC_T to_acc(R v) { return (ACC_T)v; }
template <> __device__ __forceinline__ bf16_t to_acc<bf16_t, float>(float v) {
    bf16_t r;
    r.bits = (uint16_t)(__float_as_uint(v) >> 16);            // exact: the register already holds a bf16 value
    return r;
}

// pipeline dtype of Q / K: selects the MFMA and the softmax rounding points
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct InF16 {
    static constexpr bool kBf16 = false;
    static __device__ __forceinline__ floatx4 mfma(const half8& a, const half8& b, const floatx4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
struct InBF16 {                                              // operands travel as 16 raw bytes (half8 as a bit container)
    static constexpr bool kBf16 = true;
    static __device__ __forceinline__ floatx4 mfma(const half8& a, const half8& b, const floatx4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

// f32 pair -> nearest bf16 (ties to even, one v_cvt_pk_bf16_f32): packed (low half = first element) / widened back to f32
__device__ __forceinline__ unsigned pack_bf16_pair(float2v v) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v[0]), "v"(v[1]));
    return r;
}
__device__ __forceinline__ float2v round_bf16_pair(float2v v) {
    const unsigned r = pack_bf16_pair(v);
    return float2v{__uint_as_float(r << 16), __uint_as_float(r & 0xffff0000u)};
}
// two f32 registers that hold bf16 values -> the packed pair (one v_perm_b32: the upper halves of both)
__device__ __forceinline__ unsigned pack_bf16_exact(float2v v) {
    return __builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u);
}

// C operand that starts the MFMA chain of the last token tile (mt = 4): -inf in the slots of the padding tokens 77..79 (lane quarter 3,
// r = 1..3).  Their logits then ARE -inf (finite product + -inf; every rounding keeps it; 2^-inf = 0) and the softmax needs no masking
// instructions (3 VALU per 16 pixels).  The K rows 77..79 in LDS are zero or a re-read of row 76: finite either way.
__device__ __forceinline__ floatx4 premask_tile4(int h)
{
    const float ninf = -__builtin_inff();
    return h == 3 ? floatx4{0.f, ninf, ninf, ninf} : floatx4{0.f, 0.f, 0.f, 0.f};
}

// sum of exponentials outside [2^-100, 2^100] (or inf / NaN): 1 / sum would leave the normal f32 range -> the row is redone with its maximum
__device__ __forceinline__ bool softmax_sum_out_of_range(float tot)
{
    return __float_as_uint(tot) - 0x0d800000u > 0x71800000u - 0x0d800000u;      // bit patterns of 2^-100 and 2^100; negative never occurs
}

// bf16 pipeline: logits = bf16(f32(q.k) * scale) -> f32 softmax -> bf16(p) -> acc = bf16(acc + p) (or f32 acc += p).
// Same structure as the fast fp16 path below (token 0 as the reference point, true maximum only on overflow);
// the values stay in f32 registers, so the exponent argument is a packed f32 FMA.
// softmax20_probs_bf16: the lane's 20 probabilities, rounded to bf16, as f32 pairs (shared by the tap and by daam_attend,
// whose fused sums are therefore bit-identical to the stand-alone tap's).
template <bool PREMASKED = false>
__device__ __forceinline__ void softmax20_probs_bf16(const floatx4 (&c)[5], float scale, int h, float2v (&p)[kSlots16 / 2])
{
    // scale a power of two (head_dim 64: 1/8): the multiply commutes with the bf16 rounding (bf16(c) * 2^k == bf16(c * 2^k) unless the
    // result is a bf16 subnormal), so the logits stay unscaled and the scale is folded into L -- as in the fp16 path; 10 packed
    // multiplies fewer per 16 pixels (round 5).  The conversion is the COMPILER's v_cvt_pk_bf16_f32 there (not the asm helper): it is the
    // first VALU read of the MFMA results, and only instructions the compiler can see get their MFMA -> VALU wait states padded.
    const bool pow2 = scale_is_pow2(scale);                                // wave-uniform
    float2v x[kSlots16 / 2];
    if (pow2) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned r = __builtin_bit_cast(unsigned, __builtin_convertvector(float2v{c[mt][2 * u], c[mt][2 * u + 1]}, bf16x2));
                x[2 * mt + u] = float2v{__uint_as_float(r << 16), __uint_as_float(r & 0xffff0000u)};
            }
    } else {
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            x[2 * mt] = round_bf16_pair(float2v{c[mt][0], c[mt][1]} * scale);
            x[2 * mt + 1] = round_bf16_pair(float2v{c[mt][2], c[mt][3]} * scale);
        }
    }
    if (!PREMASKED && h == 3) {                                          // tokens 77, 78, 79 (PREMASKED: -inf from the MFMA chain start)
        const float ninf = -__builtin_inff();
        x[8][1] = ninf;
        x[9] = float2v{ninf, ninf};
    }
    const float L = 1.44269502162933349609375f * (pow2 ? scale : 1.0f);   // exact: power-of-two factor
    float2v ev[kSlots16 / 2];
    auto exps = [&](float nmL) -> float {
        float2v sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < kSlots16 / 2; i += 2) {
            const float2v ta = __builtin_elementwise_fma(x[i], float2v{L, L}, float2v{nmL, nmL});
            const float2v tb = __builtin_elementwise_fma(x[i + 1], float2v{L, L}, float2v{nmL, nmL});
            ev[i] = float2v{__builtin_amdgcn_exp2f(ta[0]), __builtin_amdgcn_exp2f(ta[1])};
            ev[i + 1] = float2v{__builtin_amdgcn_exp2f(tb[0]), __builtin_amdgcn_exp2f(tb[1])};
            sa += ev[i];
            sb += ev[i + 1];
        }
        sa += sb;
        return quad_sum(sa[0] + sa[1]);
    };
    float tot = exps(0.f);                                               // reference point 0: see softmax20_probs_fast
    if (__builtin_expect(softmax_sum_out_of_range(tot), 0)) {            // huge, tiny, inf or NaN: redo with the row maximum
        float2v m2 = x[0];
#pragma unroll
        for (int i = 1; i < kSlots16 / 2; ++i) m2 = float2v{fmaxf(m2[0], x[i][0]), fmaxf(m2[1], x[i][1])};
        tot = exps(-quad_max(fmaxf(m2[0], m2[1])) * L);
    }
    const float inv = __builtin_amdgcn_rcpf(tot);
#pragma unroll
    for (int i = 0; i < kSlots16 / 2; ++i) p[i] = round_bf16_pair(ev[i] * inv);   // probs.to(dtype)
}

template <typename ACC_T, bool PREMASKED = false>
__device__ __forceinline__ void softmax20_accumulate_bf16(const floatx4 (&c)[5], const TapLayer& lay, int h,
                                                          float2v (&run)[kSlots16 / 2])
{
    float2v p[kSlots16 / 2];
    softmax20_probs_bf16<PREMASKED>(c, lay.scale, h, p);
#pragma unroll
    for (int i = 0; i < kSlots16 / 2; ++i) {
        if constexpr (sizeof(ACC_T) == 2) run[i] = round_bf16_pair(run[i] + p[i]);   // heatmap.py:156 in bf16
        else run[i] += p[i];
    }
}

// The lane's 20 fp16 probabilities (slot pairs) of one pixel, FAST flavour -- shared by the tap kernels (softmax20_accumulate) and by
// daam_attend (daam_attend_d64.hip), whose fused sums are therefore bit-identical to the stand-alone tap's.
//   logits stay packed fp16 (their reference precision); e = 2^(x L) by ONE mixed-precision FMA + v_exp_f32 per element.
//   When scale is a power of two (head_dim 64: 1/8) the multiply commutes with the fp16 rounding (fp16(c) * 2^k == fp16(c * 2^k)
//   unless the result is an fp16 subnormal, |logit| < 6.1e-5, where the two differ by < 6e-8 absolute): the logits stay unscaled
//   in fp16 and the scale is folded into L.
//   Reference point (round 4): NONE.  Softmax is shift-invariant, and cross-attention logits are small numbers (|x| <= ~40 for
//   real prompts: 2^(40 log2 e) = 2^58 against an f32 range of 2^+-126), so the exponentials are taken of the logits themselves
//   and the row is only redone with its true maximum -- as the reference does (18 v_pk_max_f16 + a lane reduction) -- when the
//   sum of a pixel leaves [2^-100, 2^100] (then 1 / sum would leave the normal range; past 2^127 the exponentials overflow, below
//   2^-126 they vanish): tests/test_gpu_parity.py::test_tap_wide_logit_spread drives both directions.  Rounds 1-3 subtracted token
//   0's logit (the start-of-text token): one f32 conversion, a five-instruction lane broadcast and a multiply per 16 pixels that buy
//   nothing.  The relative error of e is that of the one f32 rounding of x L (|x L| <= 64: <= 2.6e-6, typically 4e-7) -- the
//   same class as before and as the f32 summation order of q.k.
template <bool PREMASKED>
__device__ __forceinline__ void softmax20_probs_fast(const floatx4 (&c)[5], float scale, int h, half2v (&ph)[kSlots16 / 2])
{
    const bool pow2 = scale_is_pow2(scale);                                // wave-uniform
    half2v xh[kSlots16 / 2];
    if (pow2) {
        // compiler-generated v_cvt_pk_f16_f32 (not the asm helper): this is the first VALU read of the MFMA results,
        // and only instructions the compiler can see get their MFMA -> VALU wait states padded
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            xh[2 * mt] = __builtin_convertvector(float2v{c[mt][0], c[mt][1]}, half2v);
            xh[2 * mt + 1] = __builtin_convertvector(float2v{c[mt][2], c[mt][3]}, half2v);
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            xh[2 * mt] = cvt_pk_rne(float2v{c[mt][0], c[mt][1]} * scale);
            xh[2 * mt + 1] = cvt_pk_rne(float2v{c[mt][2], c[mt][3]} * scale);
        }
    }
    if (!PREMASKED && h == 3) {                                         // tokens 77, 78, 79 (PREMASKED: -inf from the MFMA chain start)
        const _Float16 ninf = -(_Float16)__builtin_inff();
        xh[8][1] = ninf;
        xh[9] = half2v{ninf, ninf};
    }
    const float L = 1.44269502162933349609375f * (pow2 ? scale : 1.0f);   // exact: power-of-two factor
    float2v ev[kSlots16 / 2];
    auto exps = [&](float nmL) -> float {                               // e = 2^(x L + nmL), returns the pixel's sum
        float2v sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < kSlots16 / 2; i += 2) {
            ev[i] = float2v{__builtin_amdgcn_exp2f(__builtin_fmaf((float)xh[i][0], L, nmL)),
                            __builtin_amdgcn_exp2f(__builtin_fmaf((float)xh[i][1], L, nmL))};
            ev[i + 1] = float2v{__builtin_amdgcn_exp2f(__builtin_fmaf((float)xh[i + 1][0], L, nmL)),
                                __builtin_amdgcn_exp2f(__builtin_fmaf((float)xh[i + 1][1], L, nmL))};
            sa += ev[i];
            sb += ev[i + 1];
        }
        sa += sb;
        return quad_sum(sa[0] + sa[1]);
    };
    // an addend the compiler cannot see through keeps the product a v_fma_mix_f32 straight from the fp16 logit (a plain
    // multiply would become v_cvt_f32_f16 + v_mul_f32: one instruction more per element)
    float no_shift = 0.f;
    asm("" : "+v"(no_shift));
    float tot = exps(no_shift);
    if (__builtin_expect(softmax_sum_out_of_range(tot), 0)) {           // huge, tiny, inf or NaN
        half2v ma = xh[0], mb = xh[1];
#pragma unroll
        for (int i = 2; i < kSlots16 / 2; i += 2) {
            ma = pk_max(ma, xh[i]);
            mb = pk_max(mb, xh[i + 1]);
        }
        ma = pk_max(ma, mb);
        tot = exps(-quad_max(fmaxf((float)ma[0], (float)ma[1])) * L);
    }
    const float inv = __builtin_amdgcn_rcpf(tot);                       // v_rcp_f32: 1 ulp
#pragma unroll
    for (int i = 0; i < kSlots16 / 2; ++i) ph[i] = cvt_pk_rne(ev[i] * inv);   // probs.to(dtype)
}

template <typename ACC_T, bool FAST_EXP, bool PREMASKED = false>
__device__ __forceinline__ void softmax20_accumulate(const floatx4 (&c)[5], const TapLayer& lay, int h,
                                                     typename Pair<ACC_T>::T (&run)[kSlots16 / 2])
{
    using P2 = typename Pair<ACC_T>::T;
    if constexpr (FAST_EXP) {
        half2v ph[kSlots16 / 2];
        softmax20_probs_fast<PREMASKED>(c, lay.scale, h, ph);
#pragma unroll
        for (int i = 0; i < kSlots16 / 2; ++i) run[i] += P2{(ACC_T)ph[i][0], (ACC_T)ph[i][1]};   // heatmap.py:156 (v_pk_add_f16 / v_pk_add_f32)
    } else {
        float x[kSlots16];
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = c[mt][r] * lay.scale;                    // alpha in f32, then the baddbmm output rounding
                x[4 * mt + r] = lay.round_logits ? (float)(_Float16)v : v;
            }
        if (h == 3) { x[17] = kMasked; x[18] = kMasked; x[19] = kMasked; }
        float m0 = x[0], m1 = x[1], m2 = x[2], m3 = x[3];
#pragma unroll
        for (int i = 4; i < kSlots16; i += 4) {
            m0 = fmaxf(m0, x[i]); m1 = fmaxf(m1, x[i + 1]); m2 = fmaxf(m2, x[i + 2]); m3 = fmaxf(m3, x[i + 3]);
        }
        const float m = quad_max(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)));
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < kSlots16; i += 4) {
            x[i] = exp_nonpos(x[i] - m);         s0 += x[i];
            x[i + 1] = exp_nonpos(x[i + 1] - m); s1 += x[i + 1];
            x[i + 2] = exp_nonpos(x[i + 2] - m); s2 += x[i + 2];
            x[i + 3] = exp_nonpos(x[i + 3] - m); s3 += x[i + 3];
        }
        const float inv = 1.0f / quad_sum((s0 + s1) + (s2 + s3));
#pragma unroll
        for (int i = 0; i < kSlots16; ++i) {
            const _Float16 prob = (_Float16)(x[i] * inv);                // probs.to(dtype)
            run[i >> 1][i & 1] = run[i >> 1][i & 1] + (ACC_T)prob;       // heatmap.py:156
        }
    }
}

}  // namespace daam
