// Cross-attention of one UNet layer call with the heat-map tap fused in: out = softmax(scale Q K^T) V for every
// (batch, head), and -- for the kept (conditional) heads, when the launch carries the layer's running sums -- sums += P
// from the SAME fp16 probabilities, in one kernel.  fp16 pipeline, 77 keys, head_dim a multiple of 8 up to 160 (SDXL /
// SD-2.x: 64; SD-v1.5: 40 / 80 / 160; contraction zero-padded to the next multiple of 32), gfx950, v_mfma_f32_16x16x32_f16.
//
// Replaces, inside the reference's attention processor, get_attention_scores (daam/trace.py:276; diffusers 0.21.2:
// baddbmm -> fp16 logits -> f32 softmax -> fp16 probabilities), the per-head update loop (daam/trace.py:289-294 with
// _unravel_attn :219-244 and heatmap.py:153-156) and torch.bmm(probs, value) + batch_to_head_dim (daam/trace.py:296-297),
// with the reference's rounding points:
//   logits = fp16(f32(q.k) * scale) -> f32 softmax -> p = fp16(.) -> out = fp16(sum_t f32(p_t * v_t)),  sums += p.
//
// Tiling (daam_tap16.h): workgroup = 256 threads = 4 waves = 128 pixels of one (batch, head); a wave owns 32 pixels as two
// groups of 16.  S^T = K Q^T puts the 77 probabilities of a pixel into four lanes x 20 slots; exactly that register layout
// is the B operand (P^T) of the second product O^T = V^T P^T when the key slots of V^T are stored in the same permuted
// order (v_slot_byte), so the probabilities never leave the registers between the two MFMA stages.  O^T tile: lane holds
// pixel l&15 and 4 consecutive head_dim elements -> one 8-byte store into out[batch, pixel, head*64 + ...], the
// [batch, hw, heads*64] layout the output projection consumes (no transpose / reshape copy afterwards).
// LDS (head_dim 64): K [80 rows][160 B] (A operand of S^T), V^T [64 rows][96 key slots, 208 B] (A operand of O^T), and for the
// tap a [77][128] fp16 tile of probabilities that turns the lanes' scattered 2-byte values into 16-byte row pieces of the sums.
#include "daam_tap16_softmax.h"

// (Q through full 128-byte LDS rows by LDS-DMA, as in the tap kernels, was built in round 5 and measured 5 % slower -- the call is one pass and
// latency-bound; the variant is a patch under tools/exp/patches/, LABNOTES R5.6.)

namespace daam {

constexpr int kVRow = 208;                         // bytes per V^T row: 96 key slots x 2 B + 16 pad (13 x 16 B: conflict-free b128 rows)

// byte offset of key `t` inside a V^T row: k-block kb = t / 32 of the second product, lane quarter x / 4 supplies slots
// 8*(x/4) + e, e < 4 from S^T row tile 2 kb (tokens 32 kb + ..), e >= 4 from row tile 2 kb + 1 (tokens 32 kb + 16 + ..)
__device__ __forceinline__ constexpr int v_slot_byte(int t) {
    const int kb = t >> 5, u = (t >> 4) & 1, x = t & 15;
    return (kb * 32 + (x >> 2) * 8 + u * 4 + (x & 3)) * 2;
}

// fp16 probabilities of the lane's 20 token slots (slot pairs), the arithmetic of softmax20_accumulate (daam_tap_d64.hip)
// up to the point where that one adds them to the running sums; tests/test_gpu_attend.py holds the two to bit-identical sums
template <bool FAST_EXP, bool PREMASKED>
__device__ __forceinline__ void softmax20_probs(const floatx4 (&c)[5], float scale, int round_logits, int h,
                                                half2v (&ph)[kSlots16 / 2])
{
    if constexpr (FAST_EXP) {
        if (round_logits) {                                              // wave-uniform; f32 logits (upcast_attention) take the other flavour's code
            softmax20_probs_fast<PREMASKED>(c, scale, h, ph);
            return;
        }
    }
    {
        float x[kSlots16];
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = c[mt][r] * scale;                        // alpha in f32, then the baddbmm output rounding
                x[4 * mt + r] = round_logits ? (float)(_Float16)v : v;
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
        for (int i = 0; i < kSlots16; ++i) ph[i >> 1][i & 1] = (_Float16)(x[i] * inv);   // probs.to(dtype)
    }
}

// KS = 32-wide k-steps of the first product (head_dim <= 32 KS, zero-padded), DT = 16-row tiles of the output's head_dim
// (2, 4: head_dim <= 64 -- SDXL / SD-2.x 64, SD-v1.5 40;  3, 6: <= 96 -- SD-v1.5 80;  5, 10: <= 160 -- SD-v1.5 160)
template <int KS, int DT> struct AttendShape {
    static constexpr int kKRow = KS * 64 + 32;                 // bytes per K row in LDS (+32: conflict-free b128 operand reads)
    static constexpr int kKBuf = kD64Rows * kKRow;
    static constexpr int kVBuf = DT * 16 * kVRow;
    static constexpr int kStageOff = kKBuf + kVBuf;            // probabilities tile of the tap
    static constexpr int kLds = kStageOff + kTok * kMfmaPixels * 2;
    static constexpr int kPieces = KS * 4;                     // 16-byte pieces per K / V row
    static constexpr int kKCh = (kTok * kPieces + 255) / 256;  // pieces per thread
};

// IN = InF16 / InBF16 (daam_tap16_softmax.h): the pipeline dtype selects the MFMA, the rounding points of logits, probabilities
// and output, and how the tap adds (bf16 pipelines: one softmax flavour, logits always rounded -- the host declines the rest)
template <typename IN, typename ACC_T, bool FAST_EXP, int KS, int DT>
__global__ __launch_bounds__(256, 2) void attend_kernel(const AttendLaunch L)
{
    using S = AttendShape<KS, DT>;
    constexpr int KCH = S::kKCh;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* kbuf = smem;
    unsigned char* vbuf = smem + S::kKBuf;
    _Float16* stage = reinterpret_cast<_Float16*>(smem + S::kStageOff);

    const int wg = mfma_logical_block(L.total_wgs, L.wgs_per_xcd);
    if (wg < 0) return;
    const int tid = threadIdx.x;
    const int bh = wg / L.tiles_per_head;
    const int p0 = (wg - bh * L.tiles_per_head) * kMfmaPixels;
    const int b = bh / L.heads, hd = bh - b * L.heads;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, h = lane >> 4;
    const int d = L.head_dim;                                  // multiple of 8, <= 32 KS and <= 16 DT

    // ---- global fetches first (K, V pieces; this lane's Q pieces), LDS zero-fill underneath ------------------------
    const char* kp = reinterpret_cast<const char*>(L.k) + (b * L.k_sb + hd * L.k_sh) * 2;
    const char* vp = reinterpret_cast<const char*>(L.v) + (b * L.v_sb + hd * L.v_sh) * 2;
    const char* qp = reinterpret_cast<const char*>(L.q) + (b * L.q_sb + hd * L.q_sh) * 2;
    float4v kreg[KCH], vreg[KCH];
#pragma unroll
    for (int j2 = 0; j2 < KCH; ++j2) {
        const int c = tid + 256 * j2;
        const int t = min(c / S::kPieces, kTok - 1), ch = c % S::kPieces;
        const int e = ch * 8 < d ? ch * 8 : 0;                 // pieces past head_dim: a valid address, never committed
        kreg[j2] = *as_global<float4v>(kp + ((int64_t)t * L.k_st + e) * 2);
        vreg[j2] = *as_global<float4v>(vp + ((int64_t)t * L.v_st + e) * 2);
    }
    const int px[2] = {p0 + wave * 32 + j, p0 + wave * 32 + 16 + j};
    half8 qreg[2][KS];
    {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const char* row = qp + (int64_t)min(px[g], L.hw - 1) * L.q_sp * 2;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int e = 32 * ks + 8 * h;
                qreg[g][ks] = *as_global<half8>(row + (e < d ? e : 0) * 2);
                if (e >= d) qreg[g][ks] = half8{0, 0, 0, 0, 0, 0, 0, 0};       // zero-padded contraction
            }
        }
    }
    // the tap's read-modify-write: this thread's 16-byte pieces of the running sums are fetched NOW, so that their HBM
    // latency runs under the two MFMA stages instead of behind them
    constexpr int APC = (kTok * (kMfmaPixels / 8) + 255) / 256;            // pieces per thread (5)
    constexpr int AVEC = sizeof(ACC_T) == 2 ? 1 : 2;                       // 16-byte loads per piece
    const bool tap = L.acc != nullptr && bh >= L.bh_first;                 // workgroup-uniform
    ACC_T* acc = reinterpret_cast<ACC_T*>(L.acc) + (tap ? (size_t)(bh - L.bh_first) * kTok * L.hw : 0);
    float4v areg[APC][AVEC];
#pragma unroll
    for (int a = 0; a < APC; ++a) {
        const int piece = tid + 256 * a;
        const int row = piece >> 4, col = (piece & 15) * 8;
#pragma unroll
        for (int u = 0; u < AVEC; ++u) areg[a][u] = float4v{0, 0, 0, 0};
        if (tap && !L.fresh && row < kTok && p0 + col < L.hw) {
            const ACC_T* src = acc + (size_t)row * L.hw + p0 + col;
#pragma unroll
            for (int u = 0; u < AVEC; ++u) areg[a][u] = *as_global<float4v>(src + 4 * u);
        }
    }
    // K rows 77..79 and (head_dim < 32 KS) the padding columns take part in the MFMAs: finite.  Key slots 77..95 of V^T
    // meet p = 0, rows past head_dim are computed and dropped: must not be NaN / inf either.
    if (d < 32 * KS) {
        for (int i = tid; i < S::kKBuf / 16; i += 256) *reinterpret_cast<float4v*>(kbuf + i * 16) = float4v{0, 0, 0, 0};
    } else {
        for (int i = tid; i < 3 * (S::kKRow / 16); i += 256)
            *reinterpret_cast<float4v*>(kbuf + kTok * S::kKRow + i * 16) = float4v{0, 0, 0, 0};
    }
    for (int i = tid; i < S::kVBuf / 16; i += 256) *reinterpret_cast<float4v*>(vbuf + i * 16) = float4v{0, 0, 0, 0};
    __syncthreads();
#pragma unroll
    for (int j2 = 0; j2 < KCH; ++j2) {
        const int c = tid + 256 * j2;
        const int t = c / S::kPieces, ch = c % S::kPieces;
        if (t < kTok && ch * 8 < d) {
            *reinterpret_cast<float4v*>(kbuf + t * S::kKRow + ch * 16) = kreg[j2];
            const half8 vv = __builtin_bit_cast(half8, vreg[j2]);
            unsigned char* col = vbuf + v_slot_byte(t) + (8 * ch) * kVRow;
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<_Float16*>(col + i * kVRow) = vv[i];
        }
    }
    __syncthreads();

    // ---- S^T = K Q^T ----------------------------------------------------------------------------------------------
    const unsigned char* a_rd = kbuf + j * S::kKRow + h * 16;
    floatx4 c0[5], c1[5];
    const floatx4 cmask = premask_tile4(h);                    // tokens 77..79: -inf from the start of their MFMA chain
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
        c0[mt] = mt == 4 ? cmask : floatx4{0, 0, 0, 0};
        c1[mt] = mt == 4 ? cmask : floatx4{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const half8 a = *reinterpret_cast<const half8*>(a_rd + mt * 16 * S::kKRow + ks * 64);
            c0[mt] = IN::mfma(a, qreg[0][ks], c0[mt]);
            c1[mt] = IN::mfma(a, qreg[1][ks], c1[mt]);
        }
    }
    half2v ph[2][kSlots16 / 2];                              // packed 16-bit pairs: fp16 values, or bf16 bit patterns
    if constexpr (IN::kBf16) {
        float2v pf[kSlots16 / 2];
        softmax20_probs_bf16<true>(c0, L.scale, h, pf);
#pragma unroll
        for (int i = 0; i < kSlots16 / 2; ++i) ph[0][i] = __builtin_bit_cast(half2v, pack_bf16_exact(pf[i]));
        softmax20_probs_bf16<true>(c1, L.scale, h, pf);
#pragma unroll
        for (int i = 0; i < kSlots16 / 2; ++i) ph[1][i] = __builtin_bit_cast(half2v, pack_bf16_exact(pf[i]));
    } else {
        softmax20_probs<FAST_EXP, true>(c0, L.scale, L.round_logits, h, ph[0]);
        softmax20_probs<FAST_EXP, true>(c1, L.scale, L.round_logits, h, ph[1]);
    }

    // ---- tap: probabilities of the kept heads -> LDS tile [token][pixel] ------------------------------------------
    if (tap) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < kSlots16; ++i) {
                const int t = slot16_token(i, h);
                if (t < kTok) stage[t * kMfmaPixels + wave * 32 + g * 16 + j] = ph[g][i >> 1][i & 1];
            }
    }

    // ---- O^T = V^T P^T, out[batch, pixel, head*head_dim + e] ------------------------------------------------------
    const unsigned char* v_rd = vbuf + j * kVRow + h * 16;
    _Float16* out = reinterpret_cast<_Float16*>(L.out) + b * L.o_sb + hd * L.o_sh;
    const half2v z2 = {0, 0};
    // bf16 output: the rounding instruction (v_cvt_pk_bf16_f32) only exists as inline asm, which gets no MFMA -> VALU wait
    // states from the compiler; a multiply by an OPAQUE 1.0 in front is compiler-generated code (padded) and exact
    float one = 1.0f;
    asm volatile("" : "+v"(one));
    const float2v one2 = {one, one};
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        typedef _Float16 half4v __attribute__((ext_vector_type(4)));
        half8 pb[3];
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
            const half2v e0 = ph[g][4 * kb], e1 = ph[g][4 * kb + 1];
            const half2v e2 = kb < 2 ? ph[g][(4 * kb + 2) % 10] : z2, e3 = kb < 2 ? ph[g][(4 * kb + 3) % 10] : z2;
            pb[kb] = half8{e0[0], e0[1], e1[0], e1[1], e2[0], e2[1], e3[0], e3[1]};
        }
#pragma unroll
        for (int mt = 0; mt < DT; ++mt) {
            if (16 * mt >= d) break;                           // wave-uniform
            floatx4 o = {0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
                const half8 a = *reinterpret_cast<const half8*>(v_rd + mt * 16 * kVRow + kb * 64);
                o = IN::mfma(a, pb[kb], o);
            }
            half2v lo, hi;                                     // one rounding of the f32 product sums to the pipeline dtype
            if constexpr (IN::kBf16) {
                // compiler-generated moves first: this is the first VALU read of the MFMA result (wait states)
                const float2v o01 = {o[0], o[1]}, o23 = {o[2], o[3]};
                lo = __builtin_bit_cast(half2v, pack_bf16_pair(o01 * one2));
                hi = __builtin_bit_cast(half2v, pack_bf16_pair(o23 * one2));
            } else {
                lo = __builtin_convertvector(float2v{o[0], o[1]}, half2v);
                hi = __builtin_convertvector(float2v{o[2], o[3]}, half2v);
            }
            if (px[g] < L.hw && 16 * mt + 4 * h < d)
                *as_global_rw<half4v>(out + (int64_t)px[g] * L.o_sp + 16 * mt + 4 * h) = half4v{lo[0], lo[1], hi[0], hi[1]};
        }
    }

    // ---- tap: sums[kept head][token][pixel] += p, 16-byte row pieces -----------------------------------------------
    if (tap) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < APC; ++a) {
            const int piece = tid + 256 * a;
            const int row = piece >> 4, col = (piece & 15) * 8;
            if (row >= kTok || p0 + col >= L.hw) continue;
            const half8 pv = *reinterpret_cast<const half8*>(stage + row * kMfmaPixels + col);
            ACC_T* dst = acc + (size_t)row * L.hw + p0 + col;
            if constexpr (IN::kBf16) {
                // bf16 probabilities (bit patterns in pv); sums in bf16 (heatmap.py:156 in bf16: f32 add of two bf16 values,
                // one rounding) or f32
                const ushort8 pb16 = __builtin_bit_cast(ushort8, pv);
                float pf[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) pf[i] = __uint_as_float((unsigned)pb16[i] << 16);
                if constexpr (sizeof(ACC_T) == 2) {
                    const ushort8 old = __builtin_bit_cast(ushort8, areg[a][0]);
                    ushort8 sum;
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        const unsigned r = pack_bf16_pair(float2v{__uint_as_float((unsigned)old[i] << 16) + pf[i],
                                                                  __uint_as_float((unsigned)old[i + 1] << 16) + pf[i + 1]});
                        sum[i] = (unsigned short)(r & 0xffffu);
                        sum[i + 1] = (unsigned short)(r >> 16);
                    }
                    *as_global_rw<ushort8>(dst) = sum;
                } else {
                    *as_global_rw<float4v>(dst) = areg[a][0] + float4v{pf[0], pf[1], pf[2], pf[3]};
                    *as_global_rw<float4v>(dst + 4) = areg[a][1] + float4v{pf[4], pf[5], pf[6], pf[7]};
                }
            } else if constexpr (sizeof(ACC_T) == 2) {
                half8 sum = __builtin_bit_cast(half8, areg[a][0]);
                sum += pv;                                               // heatmap.py:156 in fp16 (v_pk_add_f16)
                *as_global_rw<half8>(dst) = sum;
            } else {
                *as_global_rw<float4v>(dst) = areg[a][0] + float4v{(float)pv[0], (float)pv[1], (float)pv[2], (float)pv[3]};
                *as_global_rw<float4v>(dst + 4) = areg[a][1] + float4v{(float)pv[4], (float)pv[5], (float)pv[6], (float)pv[7]};
            }
        }
    }
}

bool attend_d64_supported(int in_dtype, int head_dim, int tokens, const int64_t* strides, int n_strides, const void* const* ptrs,
                          int n_ptrs)
{
    if ((in_dtype != 0 && in_dtype != 2) || head_dim < 8 || head_dim > 160 || head_dim % 8 != 0 || tokens != kTok) return false;
    for (int i = 0; i < n_strides; ++i)
        if (strides[i] % 8 != 0 || strides[i] < 0 || strides[i] >= ((int64_t)1 << 40)) return false;
    uintptr_t bits = 0;
    for (int i = 0; i < n_ptrs; ++i) bits |= reinterpret_cast<uintptr_t>(ptrs[i]);
    return (bits & 15) == 0;
}

template <typename IN, typename ACC_T, bool FAST, int KS, int DT>
static hipError_t launch_attend_k(const AttendLaunch& L, hipStream_t stream, int grid, int* lds_out)
{
    constexpr int lds = AttendShape<KS, DT>::kLds;
    *lds_out = lds;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attend_kernel<IN, ACC_T, FAST, KS, DT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((attend_kernel<IN, ACC_T, FAST, KS, DT>), dim3(grid), dim3(256), lds, stream, L);
    return hipGetLastError();
}

template <typename IN, typename ACC_T, bool FAST>
static hipError_t launch_attend_shape(const AttendLaunch& L, hipStream_t stream, int grid, int* lds_out)
{
    if (L.head_dim <= 64) return launch_attend_k<IN, ACC_T, FAST, 2, 4>(L, stream, grid, lds_out);
    if (L.head_dim <= 96) return launch_attend_k<IN, ACC_T, FAST, 3, 6>(L, stream, grid, lds_out);
    return launch_attend_k<IN, ACC_T, FAST, 5, 10>(L, stream, grid, lds_out);
}

hipError_t launch_attend_d64(const AttendLaunch& L, int in_dtype, int acc_dtype, int fast_exp, hipStream_t stream, int* grid_out, int* lds_out)
{
    const int grid = L.wgs_per_xcd * 8;
    *grid_out = grid;
    if (in_dtype == 2) {                                       // bf16 pipeline: bf16 or f32 sums, one softmax flavour
        if (acc_dtype == 2) return launch_attend_shape<InBF16, bf16_t, true>(L, stream, grid, lds_out);
        if (acc_dtype == 1) return launch_attend_shape<InBF16, float, true>(L, stream, grid, lds_out);
        return hipErrorInvalidValue;
    }
    if (acc_dtype == 0)
        return fast_exp ? launch_attend_shape<InF16, _Float16, true>(L, stream, grid, lds_out) : launch_attend_shape<InF16, _Float16, false>(L, stream, grid, lds_out);
    if (acc_dtype == 1)
        return fast_exp ? launch_attend_shape<InF16, float, true>(L, stream, grid, lds_out) : launch_attend_shape<InF16, float, false>(L, stream, grid, lds_out);
    return hipErrorInvalidValue;
}

}  // namespace daam
