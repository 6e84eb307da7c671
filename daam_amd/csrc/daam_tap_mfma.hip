// fp16 tap for gfx950: S^T = K Q^T on the matrix cores, softmax in registers, running sums
// in registers across every step of the launch, one coalesced write-back per launch.
//
// Reference semantics (daam/trace.py:276-294, daam/heatmap.py:153-156, diffusers 0.21.2
// Attention.get_attention_scores):
//   logits = fp16( f32(q . k) * scale )            (baddbmm output dtype)
//   probs  = fp16( softmax_f32(logits) )           (probs.to(dtype))
//   acc    = acc + probs  in the accumulator dtype (fp16 add = f32 add + one RNE)
//
// Workgroup = 256 threads = 4 waves, one (layer, kept head, 128-pixel tile).  Wave w owns
// pixels [32w, 32w+32).  "Swapped" product: A = K (tokens x d), B = Q^T (d x pixels) with
// v_mfma_f32_32x32x16_f16, so a lane holds ONE pixel (column lane&31) and, over the three
// 32-token row tiles, 40 of its 77 token logits; the row reduction of the softmax is 39 in-lane
// ops + one exchange with lane^32.  The contraction index is split as (k-step ks, lane half
// g, element e) <-> d index 16*ks + 8*g + e for both operands, i.e. every operand fetch is
// one aligned 16-byte piece of a q / k row.
//
// K of the current step sits in LDS (rows padded by 16 B: stride 4*(2*KS+1) dwords, odd
// multiple of 4 -> ds_read_b128 conflict-free), double-buffered; Q and the next step's K are
// fetched global -> VGPR one step ahead.
#include "daam_tap_common.h"

namespace daam {

#define DAAM_T(i) do {} while (0)

template <int KS, typename ACC_T>
constexpr size_t tap_mfma_lds_bytes() {
    // K double buffer and the write-back staging tile are never live together: aliased
    const size_t kb = 2 * (size_t)kTokRows * (KS * 32 + 16), st = (size_t)kTok * kMfmaPixels * sizeof(ACC_T);
    return (kb > st ? kb : st) + (size_t)kMaxStepsPerLaunch * 2 * sizeof(void*);
}

// waves per SIMD the register allocator must leave room for: 4 for the fp16-sum SD/SDXL head dims
template <int KS, typename ACC_T, bool FAST> constexpr int tap_mfma_min_waves() {
    return (KS <= 4 && sizeof(ACC_T) == 2) ? (FAST ? 3 : 4) : 2;
}

template <int KS, typename ACC_T, bool FAST_EXP>
__global__ __launch_bounds__(256, (tap_mfma_min_waves<KS, ACC_T, FAST_EXP>())) void tap_mfma_kernel(const TapLaunch L)
{
    constexpr int KROW = KS * 32 + 16;                        // bytes per K row in LDS
    constexpr int KBUF = kTokRows * KROW;                     // bytes per K buffer
    constexpr int KCH = (kTok * 2 * KS + 255) / 256;          // 16-B K pieces per thread per step
    constexpr int VEC = AccVec<ACC_T>::kPerVec;
    constexpr int PPR = kMfmaPixels / VEC;                    // 16-B pieces per staging row
    constexpr size_t kPtrOff = tap_mfma_lds_bytes<KS, ACC_T>() - (size_t)kMaxStepsPerLaunch * 2 * sizeof(void*);

    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* kbuf = smem;                               // [2][KBUF]
    ACC_T* stage = reinterpret_cast<ACC_T*>(smem);            // [kTok][kMfmaPixels], aliases kbuf
    const void** sptr = reinterpret_cast<const void**>(smem + kPtrOff);

    const int wg = mfma_logical_block(L.total_wgs, L.wgs_per_xcd);
    if (wg < 0) return;
    tap_mark_started(L);
    TapLayer lay;
    const bool table = L.layers != nullptr;
    if (table) {
        const DAAM_GLOBAL TapLayer* gl = as_global<TapLayer>(L.layers);
        load_layer(gl + mfma_find_layer(gl, L.n_layers, wg), &lay);
    } else {
        lay = L.one;
    }
    const int tid = threadIdx.x;
    // per-step q / k base pointers -> LDS once, so the step loop never waits on a dependent
    // global load (table fetch -> address -> data) on its critical path
    if (table) {
        const DAAM_GLOBAL TapPtr* ptrs = as_global<TapPtr>(L.ptrs) + lay.ptr_begin;
        for (int i = tid; i < lay.n_steps; i += 256) {
            sptr[2 * i] = ptrs[i].q;
            sptr[2 * i + 1] = ptrs[i].k;
        }
    } else if (tid == 0) {
        sptr[0] = L.one_ptr.q;
        sptr[1] = L.one_ptr.k;
    }
    const int n_steps = lay.n_steps;
    const int nch = lay.head_dim >> 3;                        // 16-B pieces per q / k row
    const int rel = wg - lay.wg_begin;
    const int kh = rel / lay.tiles_per_head;
    const int p0 = (rel - kh * lay.tiles_per_head) * kMfmaPixels;
    const int bh = lay.bh_first + kh;
    const int b = bh / lay.heads, h = bh - b * lay.heads;
    const int64_t k_off = b * lay.k_sb + h * lay.k_sh;

    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, g = lane >> 5;
    const int my_pixel = min(p0 + wave * 32 + n, lay.hw - 1);  // clamped: out-of-range columns are never stored
    const int64_t q_row = b * lay.q_sb + h * lay.q_sh + (int64_t)my_pixel * lay.q_sp;

    // ---- running sums -> registers (through the staging tile, 16-byte row pieces) --------------
    ACC_T run[kSlots];
    ACC_T* acc = reinterpret_cast<ACC_T*>(lay.acc) + (size_t)kh * kTok * lay.hw;
    if (!lay.fresh) {
        for (int piece = tid; piece < kTok * PPR; piece += 256) {
            const int row = piece / PPR, col = (piece - row * PPR) * VEC;
            if (p0 + col < lay.hw)
                *reinterpret_cast<float4v*>(stage + row * kMfmaPixels + col) =
                    *as_global<float4v>(acc + (size_t)row * lay.hw + p0 + col);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kSlots; ++i) {
            const int t = slot_token(i, g);
            run[i] = t < kTok ? stage[t * kMfmaPixels + wave * 32 + n] : (ACC_T)0;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kSlots; ++i) run[i] = (ACC_T)0;
    }
    __syncthreads();                                          // staging reads done; sptr visible
    // zero the K padding pieces (head_dim not a multiple of 16: piece 2*KS-1) once, both buffers
    if (nch < 2 * KS) {
        for (int r = tid; r < 2 * kTokRows; r += 256)
            *reinterpret_cast<float4v*>(kbuf + (r / kTokRows) * KBUF + (r % kTokRows) * KROW + nch * 16) = float4v{0, 0, 0, 0};
    }

    // per-thread K piece coordinates (fixed for the whole launch); rows >= 77 are clamped duplicates
    int k_src[KCH], k_dst[KCH];
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
        const int c = tid + 256 * j;
        const int t = c / nch, ch = c - t * nch;
        k_src[j] = min(t, kTok - 1) * (int)lay.k_st + ch * 8;
        k_dst[j] = t < kTok ? t * KROW + ch * 16 : -1;
    }
    float4v kreg[KCH];
    half8 bq[KS];
    auto issue_k = [&](int s) {
        const _Float16* kp = reinterpret_cast<const _Float16*>(sptr[2 * s + 1]) + k_off;
#pragma unroll
        for (int j = 0; j < KCH; ++j) kreg[j] = *as_global<float4v>(kp + k_src[j]);
    };
    auto commit_k = [&](int buf) {
#pragma unroll
        for (int j = 0; j < KCH; ++j)
            if (k_dst[j] >= 0) *reinterpret_cast<float4v*>(kbuf + buf * KBUF + k_dst[j]) = kreg[j];
    };
    auto issue_q = [&](int s) {
        const _Float16* qp = reinterpret_cast<const _Float16*>(sptr[2 * s]) + q_row;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bq[ks] = *as_global<half8>(qp + min(2 * ks + g, nch - 1) * 8);
        // head_dim not a multiple of 16: the upper half's last piece is a clamped DUPLICATE of real
        // data; its partner A piece is zero in LDS, so it contributes exactly 0.  (Zeroing it here
        // would put a use of the freshly issued loads at the end of every step.)
    };

    // ---- step loop: one barrier per step ------------------------------------------------------
    //   top:  barrier (K(s) committed by everyone; everyone is done reading the other buffer)
    //   MFMA: S^T = K(s) Q(s)^T      (operands: LDS buffer s&1, bq)
    //   then: issue the global loads of step s+1 (K -> kreg, Q -> bq; the operand registers are free
    //         again) so that they fly under the softmax, the longest phase
    //   softmax + accumulate in registers
    //   end:  kreg -> LDS buffer (s+1)&1
    issue_k(0);
    issue_q(0);
    commit_k(0);
    for (int s = 0; s < n_steps; ++s) {
        DAAM_T(0);                                            // commit_k + loop overhead of the previous step
        __syncthreads();
        DAAM_T(1);                                            // barrier wait
        const unsigned char* kb = kbuf + (s & 1) * KBUF;
        floatx16 c0 = {0}, c1 = {0}, c2 = {0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int col = (2 * ks + g) * 16;
            const half8 a0 = *reinterpret_cast<const half8*>(kb + (n) * KROW + col);
            const half8 a1 = *reinterpret_cast<const half8*>(kb + (32 + n) * KROW + col);
            const half8 a2 = *reinterpret_cast<const half8*>(kb + (64 + n) * KROW + col);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bq[ks], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bq[ks], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, bq[ks], c2, 0, 0, 0);
        }
        asm volatile("" :: "v"(c0[0]), "v"(c1[0]), "v"(c2[0]));
        DAAM_T(2);                                            // LDS operand reads + 12 MFMA
        const int nx = min(s + 1, n_steps - 1);               // branch-free: the last step re-fetches itself
        issue_k(nx);
        issue_q(nx);

        softmax_accumulate<ACC_T, FAST_EXP>(c0, c1, c2, lay, g, run);
        asm volatile("" :: "v"(run[0]), "v"(run[39]));
        DAAM_T(3);                                            // load issue + softmax + accumulate
        commit_k((s + 1) & 1);
    }
    __syncthreads();                                          // all K reads done before the staging tile reuses the space

    // ---- write back: registers -> LDS [token][pixel] -> 16-byte row pieces -------------------
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        const int t = slot_token(i, g);
        if (t < kTok) stage[t * kMfmaPixels + wave * 32 + n] = run[i];
    }
    __syncthreads();
    for (int piece = tid; piece < kTok * PPR; piece += 256) {
        const int row = piece / PPR, col = (piece - row * PPR) * VEC;
        if (p0 + col < lay.hw)
            *as_global_rw<float4v>(acc + (size_t)row * lay.hw + p0 + col) =
                *reinterpret_cast<const float4v*>(stage + row * kMfmaPixels + col);
    }
}

// ---------------------------------------------------------------------------------------
bool tap_mfma_supported(int in_dtype, int head_dim, int tokens, int hw, int64_t q_sp, int64_t k_st, int64_t q_sb,
                        int64_t q_sh, int64_t k_sb, int64_t k_sh)
{
    if (in_dtype != 0 || tokens != kTok) return false;
    if (head_dim % 8 != 0 || head_dim < 8 || head_dim > 160) return false;
    if (hw % 8 != 0) return false;
    const int64_t s[] = {q_sp, k_st, q_sb, q_sh, k_sb, k_sh};
    for (int64_t v : s)
        if (v % 8 != 0) return false;                       // every row piece 16-byte aligned
    return true;
}

int tap_mfma_tile_pixels() { return kMfmaPixels; }
int tap_mfma_max_steps() { return kMaxStepsPerLaunch; }
int tap_mfma_ksteps(int head_dim) { return (head_dim + 15) / 16; }

template <int KS, typename ACC_T, bool FAST>
static hipError_t launch_one(const TapLaunch& L, hipStream_t stream, int grid, size_t* lds_out)
{
    const size_t lds = tap_mfma_lds_bytes<KS, ACC_T>();
    *lds_out = lds;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tap_mfma_kernel<KS, ACC_T, FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((tap_mfma_kernel<KS, ACC_T, FAST>), dim3(grid), dim3(256), lds, stream, L);
    return hipGetLastError();
}

// every layer of the launch has the same k-step count ceil(head_dim / 16) (host groups by it)
hipError_t launch_tap_mfma(const TapLaunch& L, int acc_dtype, int max_d, int fast_exp, hipStream_t stream,
                           int* grid_out, int* lds_out)
{
    const int grid = L.wgs_per_xcd * 8;
    *grid_out = grid;
    size_t lds = 0;
    hipError_t e = hipErrorInvalidValue;
    const int ks = tap_mfma_ksteps(max_d);
#define DAAM_CASE(K)                                                                                  \
    case K:                                                                                           \
        if (fast_exp) e = acc_dtype == 0 ? launch_one<K, _Float16, true>(L, stream, grid, &lds)       \
                                         : launch_one<K, float, true>(L, stream, grid, &lds);         \
        else e = acc_dtype == 0 ? launch_one<K, _Float16, false>(L, stream, grid, &lds)               \
                                : launch_one<K, float, false>(L, stream, grid, &lds);                 \
        break;
    switch (ks) {
        DAAM_CASE(1) DAAM_CASE(2) DAAM_CASE(3) DAAM_CASE(4) DAAM_CASE(5) DAAM_CASE(6) DAAM_CASE(7) DAAM_CASE(8)
        DAAM_CASE(9) DAAM_CASE(10)
        default: break;
    }
#undef DAAM_CASE
    *lds_out = (int)lds;
    return e;
}

}  // namespace daam
