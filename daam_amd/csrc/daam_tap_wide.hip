// fp16 tap for 64 < head_dim <= 160 (SD-v1.5: 80 on its 32x32 layers, 160 on its 16x16 layers; SD-2.x 768-v has 64
// everywhere), gfx950: the 16x16x32 tiling and the full-row LDS data path of daam_tap_d64.hip with KS = 3 or 5 k-steps of
// 32 instead of 2.
//
// Why not the 32x32x16 kernel (daam_tap_mfma.hip) these layers ran on in round 1: they are SMALL -- 8 heads x 8 or 2 tiles
// x 5 layers = 320 / 80 workgroups for a whole 50-step launch -- so what a launch costs is one workgroup's chain of 50
// dependent steps, with nothing else on its CU to hide a stall.  There the per-step latency is everything: the 32x32 kernel
// fetched Q in MFMA layout (32 rows x 32 B per load instruction) and spent 7-9 us per step.  Here K and Q come in full rows by
// coalesced buffer loads one step ahead, cross LDS once, and a step is 10 KS MFMAs + one softmax per 16-pixel group.
//
// Same arithmetic and rounding points as the other tap kernels (daam_tap16_softmax.h); with the same tiling as daam_tap_d64
// and daam_attend_d64 the three leave bit-identical sums for the same inputs.
// LDS rows are head_dim-independent per instantiation: KS * 64 bytes of data + 16 bytes of padding (13 or 21 sixteen-byte
// units: an odd count makes the operand reads -- 16 rows x one chunk per ds_read_b128 -- conflict-free); chunks past head_dim
// (80 = 10 of 12 chunks) stay zero.  K double-buffered, Q in a wave-private tile, register-staged one step ahead.
#include "daam_tap16_softmax.h"

namespace daam {

template <int KS> struct WideShape {
    static constexpr int kRow = KS * 64 + 16;                  // bytes per K / Q row in LDS
    static constexpr int kChunks = KS * 4;                     // 16-byte chunk slots per row
    static constexpr int kKBuf = kD64Rows * kRow;              // 80 rows, 77..79 stay zero
    static constexpr int kQTile = 32 * kRow;                   // one wave's 32 pixel rows
    static constexpr int kQOff = 2 * kKBuf;
    static constexpr int kKCh = (kTok * kChunks + 255) / 256;  // K pieces per thread per step
    static constexpr int kQCh = 32 * kChunks / 64;             // Q pieces per lane per step
};

template <typename ACC_T, int KS> constexpr size_t tap_wide_lds_bytes() {
    using S = WideShape<KS>;
    const size_t kb = (size_t)S::kQOff + 4 * (size_t)S::kQTile, st = (size_t)kTok * kMfmaPixels * sizeof(ACC_T);
    return (kb > st ? kb : st) + (size_t)kMaxStepsPerLaunch * 2 * sizeof(void*);
}

template <typename ACC_T, bool FAST_EXP, int KS>
__global__ __launch_bounds__(256, (KS > 3 ? 1 : 2)) void tap_wide_kernel(const TapLaunch L)   // KS 5: 98 KB of LDS, one workgroup per CU anyway
{
    using S = WideShape<KS>;
    constexpr int KCH = S::kKCh, QCH = S::kQCh;
    constexpr int VEC = AccVec<ACC_T>::kPerVec;
    constexpr int PPR = kMfmaPixels / VEC;
    constexpr size_t kPtrOff = tap_wide_lds_bytes<ACC_T, KS>() - (size_t)kMaxStepsPerLaunch * 2 * sizeof(void*);

    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* kbuf = smem;                               // [2][kKBuf], then the four waves' Q tiles
    ACC_T* stage = reinterpret_cast<ACC_T*>(smem);            // [kTok][kMfmaPixels], aliases both
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
    const int rel = wg - lay.wg_begin;
    const int kh = rel / lay.tiles_per_head;
    const int p0 = (rel - kh * lay.tiles_per_head) * kMfmaPixels;
    const int bh = lay.bh_first + kh;
    const int b = bh / lay.heads, hd = bh - b * lay.heads;
    const int64_t k_off = b * lay.k_sb + hd * lay.k_sh;
    const int64_t q_off = b * lay.q_sb + hd * lay.q_sh;

    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, h = lane >> 4;

    // ---- running sums -> registers (through the staging tile, 16-byte row pieces) --------------
    typename Pair<ACC_T>::T run0[kSlots16 / 2], run1[kSlots16 / 2];   // slot pairs (2i, 2i+1)
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
        for (int i = 0; i < kSlots16; ++i) {
            const int t = slot16_token(i, h);
            if (t < kTok) {
                run0[i >> 1][i & 1] = from_acc<ACC_T>(stage[t * kMfmaPixels + wave * 32 + j]);
                run1[i >> 1][i & 1] = from_acc<ACC_T>(stage[t * kMfmaPixels + wave * 32 + 16 + j]);
            } else {
                run0[i >> 1][i & 1] = 0;
                run1[i >> 1][i & 1] = 0;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < kSlots16; ++i) { run0[i >> 1][i & 1] = 0; run1[i >> 1][i & 1] = 0; }
    }
    __syncthreads();                                          // staging reads done; sptr visible
    // K rows 77..79 and the chunk slots past head_dim take part in the MFMAs: both K buffers and the Q tiles start as zeros,
    // the steps only ever write the chunks inside head_dim
    for (int i = tid; i < (S::kQOff + 4 * S::kQTile) / 16; i += 256)
        *reinterpret_cast<float4v*>(kbuf + i * 16) = float4v{0, 0, 0, 0};
    __syncthreads();

    const int d = lay.head_dim;                               // multiple of 8, <= 32 KS
    // K pieces of this thread: piece c = tid + 256 i -> row c / kChunks, chunk c % kChunks
    unsigned k_src[KCH];
    int k_dst[KCH];
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const int c = tid + 256 * i;
        const int t = c / S::kChunks, ch = c % S::kChunks;
        const bool in = t < kTok && ch * 8 < d;
        k_src[i] = (unsigned)((min(t, kTok - 1) * (int)lay.k_st + (ch * 8 < d ? ch * 8 : 0)) * 2);
        k_dst[i] = in ? t * S::kRow + ch * 16 : -1;
    }
    // Q pieces of this lane: piece p = lane + 64 i of the wave's 32 pixel rows -> row p / kChunks, chunk p % kChunks
    // (kChunks consecutive lanes fetch one whole row)
    unsigned q_src[QCH];
    int q_dst[QCH];
#pragma unroll
    for (int i = 0; i < QCH; ++i) {
        const int p = lane + 64 * i;
        const int row = p / S::kChunks, ch = p % S::kChunks;
        const int px = min(p0 + wave * 32 + row, lay.hw - 1);
        q_src[i] = (unsigned)((q_off + (int64_t)px * lay.q_sp + (ch * 8 < d ? ch * 8 : 0)) * 2);
        q_dst[i] = ch * 8 < d ? row * S::kRow + ch * 16 : -1;
    }
    unsigned char* qtile = kbuf + S::kQOff + wave * S::kQTile;
    const int f_rd = j * S::kRow + h * 16;                    // operand reads: row l&15 of a 16-row tile, chunk 4 ks + (l >> 4)

    auto tensor = [](const void* p) -> __amdgpu_buffer_rsrc_t {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, -1, 0x00020000);
    };
    const unsigned k_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(k_off * 2));

    float4v kreg[KCH], qreg[QCH];
    auto issue = [&](int s) {
        const __amdgpu_buffer_rsrc_t kt = tensor(sptr[2 * s + 1]), qt = tensor(sptr[2 * s]);
#pragma unroll
        for (int i = 0; i < KCH; ++i) kreg[i] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(kt, k_src[i], k_base, 0));
#pragma unroll
        for (int i = 0; i < QCH; ++i) qreg[i] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(qt, q_src[i], 0, 0));
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < KCH; ++i)
            if (k_dst[i] >= 0) *reinterpret_cast<float4v*>(kbuf + buf * S::kKBuf + k_dst[i]) = kreg[i];
#pragma unroll
        for (int i = 0; i < QCH; ++i)
            if (q_dst[i] >= 0) *reinterpret_cast<float4v*>(qtile + q_dst[i]) = qreg[i];
    };

    issue(0);
    commit(0);
    const floatx4 cmask = premask_tile4(h);
    for (int s = 0; s < n_steps; ++s) {
        __syncthreads();
        const unsigned char* kb = kbuf + (s & 1) * S::kKBuf + f_rd;
        floatx4 c0[5], c1[5];
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            c0[mt] = mt == 4 ? cmask : floatx4{0, 0, 0, 0};     // tokens 77..79: -inf from the start of their MFMA chain
            c1[mt] = mt == 4 ? cmask : floatx4{0, 0, 0, 0};
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const half8 q0 = *reinterpret_cast<const half8*>(qtile + f_rd + ks * 64);
            const half8 q1 = *reinterpret_cast<const half8*>(qtile + 16 * S::kRow + f_rd + ks * 64);
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                const half8 a = *reinterpret_cast<const half8*>(kb + mt * 16 * S::kRow + ks * 64);
                c0[mt] = InF16::mfma(a, q0, c0[mt]);
                c1[mt] = InF16::mfma(a, q1, c1[mt]);
            }
        }
        issue(min(s + 1, n_steps - 1));                       // branch-free: the last step re-fetches itself
        softmax20_accumulate<ACC_T, FAST_EXP, true>(c0, lay, h, run0);
        softmax20_accumulate<ACC_T, FAST_EXP, true>(c1, lay, h, run1);
        commit((s + 1) & 1);
    }
    __syncthreads();                                          // all operand reads done before the staging tile reuses the space

    // ---- write back: registers -> LDS [token][pixel] -> 16-byte row pieces -------------------
#pragma unroll
    for (int i = 0; i < kSlots16; ++i) {
        const int t = slot16_token(i, h);
        if (t < kTok) {
            stage[t * kMfmaPixels + wave * 32 + j] = to_acc<ACC_T>(run0[i >> 1][i & 1]);
            stage[t * kMfmaPixels + wave * 32 + 16 + j] = to_acc<ACC_T>(run1[i >> 1][i & 1]);
        }
    }
    __syncthreads();
    for (int piece = tid; piece < kTok * PPR; piece += 256) {
        const int row = piece / PPR, col = (piece - row * PPR) * VEC;
        if (p0 + col < lay.hw)
            *as_global_rw<float4v>(acc + (size_t)row * lay.hw + p0 + col) =
                *reinterpret_cast<const float4v*>(stage + row * kMfmaPixels + col);
    }
}

bool tap_wide_supported(int in_dtype, int head_dim, int hw, int64_t q_sp, int64_t k_st, int64_t q_sb, int64_t q_sh, int64_t k_sb,
                        int64_t k_sh, const void* q, const void* k)
{
    if (in_dtype != 0 || head_dim <= 64 || head_dim > 160 || head_dim % 8 != 0 || hw % 8 != 0) return false;
    const int64_t s[] = {q_sp, k_st, q_sb, q_sh, k_sb, k_sh};
    for (int64_t v : s)
        if (v % 8 != 0) return false;
    if (k_st * 77 >= (int64_t)1 << 30 || q_sp * (int64_t)hw >= (int64_t)1 << 30) return false;   // byte offsets stay in 32 bits
    return ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15) == 0;
}

template <typename ACC_T, bool FAST, int KS>
static hipError_t launch_wide_k(const TapLaunch& L, hipStream_t stream, int grid, size_t* lds_out)
{
    const size_t lds = tap_wide_lds_bytes<ACC_T, KS>();
    *lds_out = lds;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tap_wide_kernel<ACC_T, FAST, KS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((tap_wide_kernel<ACC_T, FAST, KS>), dim3(grid), dim3(256), lds, stream, L);
    return hipGetLastError();
}

// max_head_dim of the launch's layers selects the shape: <= 96 -> KS 3, <= 160 -> KS 5
hipError_t launch_tap_wide(const TapLaunch& L, int acc_dtype, int max_head_dim, int fast_exp, hipStream_t stream, int* grid_out, int* lds_out)
{
    const int grid = L.wgs_per_xcd * 8;
    *grid_out = grid;
    size_t lds = 0;
    hipError_t e;
    const bool ks3 = max_head_dim <= 96;
    if (acc_dtype == 0) {
        if (fast_exp) e = ks3 ? launch_wide_k<_Float16, true, 3>(L, stream, grid, &lds) : launch_wide_k<_Float16, true, 5>(L, stream, grid, &lds);
        else e = ks3 ? launch_wide_k<_Float16, false, 3>(L, stream, grid, &lds) : launch_wide_k<_Float16, false, 5>(L, stream, grid, &lds);
    } else if (acc_dtype == 1) {
        if (fast_exp) e = ks3 ? launch_wide_k<float, true, 3>(L, stream, grid, &lds) : launch_wide_k<float, true, 5>(L, stream, grid, &lds);
        else e = ks3 ? launch_wide_k<float, false, 3>(L, stream, grid, &lds) : launch_wide_k<float, false, 5>(L, stream, grid, &lds);
    } else {
        return hipErrorInvalidValue;
    }
    *lds_out = (int)lds;
    return e;
}

}  // namespace daam
