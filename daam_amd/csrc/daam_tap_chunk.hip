// fp16 (and, opt-in, bf16) tap for ANY head_dim that is a multiple of 8 up to 256, in the LDS / register footprint of the head_dim-64 kernel
// (daam_tap_d64.hip): the contraction is walked in CHUNKS of 64 elements.  gfx950, v_mfma_f32_16x16x32_f16.
//
// Why (SD-v1.5, BASELINE.json configs[1]): its layers have head_dim 40 / 80 / 160, and a flush ran as three kernels side by
// side -- tap_d64_kernel (1280 workgroups, 37.9 KB of LDS) and tap_wide_kernel<3|5> (320 / 80 workgroups, 61 / 98 KB: full
// rows of 160 or 320 bytes in LDS).  The wide kernels' LDS is what serialises the three (a CU takes one
// head_dim-40 workgroup instead of four beside a wide one) and the head_dim-40 grid alone is 1.25 rounds of the chip.  Here
// a row tile of K / Q in LDS is always 128 bytes wide: a step of a head_dim-160 layer is three sub-steps (64 + 64 + 32
// elements) whose MFMAs accumulate into the same C registers, the softmax runs after the last one.  Every layer of a
// flush, whatever its head_dim, is then ONE kind of workgroup: one launch, four workgroups per CU, no side streams, no
// start gate.
//
// Same tiling, operand layout and k order as daam_tap_d64.hip / daam_tap_wide.hip (k-step of 32 elements number 2 c + ks of
// chunk c; lane quarter h contracts elements 32 (2 c + ks) + 8 h .. + 7), same softmax (daam_tap16_softmax.h): the sums
// it leaves are bit-identical to theirs.
//
// Data path: both operands travel HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), swizzle applied to the source address,
// one sub-step ahead (K double-buffered, Q in a wave-private tile, as in the head_dim-64 kernel).  A partial last chunk
// (head_dim 40: 5 of 8 sixteen-byte pieces, 80: 2, 160: 4) cannot be zero-padded by a DMA: the lanes whose piece lies past
// head_dim re-fetch piece 0 of their row (a valid address, finite data) and the Q operand of the k-step is cleared in
// registers after the LDS read (K x 0 = 0); a k-step that lies entirely past head_dim is skipped.
#include "daam_tap16_softmax.h"

namespace daam {

constexpr int kCkRow = 128;                         // bytes per K / Q row in LDS: one 64-element chunk, 16-byte pieces swizzled
constexpr int kCkKBuf = kD64Rows * kCkRow;          // 10240: 80 K rows (77..79 re-read row 76: finite, their logits are masked)
constexpr int kCkQTile = 32 * kCkRow;               // 4096: one wave's 32 pixel rows
constexpr int kCkQOff = 2 * kCkKBuf;                // Q tiles of the four waves follow the two K buffers
constexpr int kCkMaxHeadDim = 256;

template <typename ACC_T> constexpr size_t tap_chunk_lds_bytes() {
    const size_t kb = 2 * (size_t)kCkKBuf + 4 * (size_t)kCkQTile, st = (size_t)kTok * kMfmaPixels * sizeof(ACC_T);
    return (kb > st ? kb : st) + (size_t)kMaxStepsPerLaunch * 2 * sizeof(void*);     // fp16 sums: 37888 -> 4 workgroups per CU
}

__device__ __forceinline__ constexpr int ck_swz(int row, int chunk) { return ((chunk ^ ((row >> 1) & 7)) << 4); }

// The MFMAs of one sub-step: C (+)= K_chunk Q_chunk^T for the wave's two 16-pixel groups.  `partial`: the chunk is the layer's
// last and holds only `vc` (1..7) valid 16-byte pieces.  All conditions but the lane-quarter compares are wave-uniform.
// (`partial` as a second template parameter -- full chunks without the selects -- costs registers: 128 VGPRs + 160 bytes of
// scratch against 125 and none; hipcc turns the two `if`s below into eight v_cndmask each either way.)
template <typename IN, bool FIRST>
__device__ __forceinline__ void chunk_mfma(const unsigned char* kb, const unsigned char* qtile, int f_rd, bool partial, int vc, int h,
                                           floatx4 (&c0)[5], floatx4 (&c1)[5], const floatx4& cmask)
{
    const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    half8 q00 = *reinterpret_cast<const half8*>(qtile + f_rd);
    half8 q10 = *reinterpret_cast<const half8*>(qtile + 16 * kCkRow + f_rd);
    if (partial && vc < 4 && h >= vc) { q00 = zero; q10 = zero; }          // k-step 0 = pieces 0..3: lane quarter h holds piece h
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
        const half8 a0 = *reinterpret_cast<const half8*>(kb + mt * 16 * kCkRow + f_rd);
        if constexpr (FIRST) {                                             // tokens 77..79 (tile 4): -inf from the start of their chain
            c0[mt] = IN::mfma(a0, q00, mt == 4 ? cmask : floatx4{0, 0, 0, 0});
            c1[mt] = IN::mfma(a0, q10, mt == 4 ? cmask : floatx4{0, 0, 0, 0});
        } else {
            c0[mt] = IN::mfma(a0, q00, c0[mt]);
            c1[mt] = IN::mfma(a0, q10, c1[mt]);
        }
    }
    if (!partial || vc > 4) {                                              // k-step 1 = pieces 4..7: piece 4 + h
        half8 q01 = *reinterpret_cast<const half8*>(qtile + (f_rd ^ 64));
        half8 q11 = *reinterpret_cast<const half8*>(qtile + 16 * kCkRow + (f_rd ^ 64));
        if (partial && 4 + h >= vc) { q01 = zero; q11 = zero; }
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            const half8 a1 = *reinterpret_cast<const half8*>(kb + mt * 16 * kCkRow + (f_rd ^ 64));
            c0[mt] = IN::mfma(a1, q01, c0[mt]);
            c1[mt] = IN::mfma(a1, q11, c1[mt]);
        }
    }
}

// IN = InF16 / InBF16 (daam_tap16_softmax.h: the MFMA and the softmax rounding points of the pipeline dtype; bf16 has one softmax
// flavour and keeps its values in f32 registers: 3 waves per SIMD)
template <typename IN, typename ACC_T, bool FAST_EXP>
__global__ __launch_bounds__(256, ((sizeof(ACC_T) == 2 && !IN::kBf16) ? 4 : 3)) void tap_chunk_kernel(const TapLaunch L)
{
    constexpr int VEC = AccVec<ACC_T>::kPerVec;
    constexpr int PPR = kMfmaPixels / VEC;
    constexpr size_t kPtrOff = tap_chunk_lds_bytes<ACC_T>() - (size_t)kMaxStepsPerLaunch * 2 * sizeof(void*);

    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* kbuf = smem;                               // [2][kCkKBuf], then the four waves' Q tiles
    ACC_T* stage = reinterpret_cast<ACC_T*>(smem);            // [kTok][kMfmaPixels], aliases both
    const void** sptr = reinterpret_cast<const void**>(smem + kPtrOff);

    // blockIdx -> logical workgroup.  wgs_per_xcd > 0: every XCD takes a contiguous range (the tiles of a head share K in one L2;
    // launches of one head_dim).  wgs_per_xcd < 0 (launch_tap_chunk, mixed head dims): consecutive logical workgroups go to
    // consecutive XCDs, so every XCD gets the same mix of light (head_dim <= 64: one sub-step per step) and heavy tiles.
    int wg;
    if (L.wgs_per_xcd > 0) wg = mfma_logical_block(L.total_wgs, L.wgs_per_xcd);
    else wg = (int)blockIdx.x < L.total_wgs ? (int)blockIdx.x : -1;
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

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the DMA block choice must not become exec masks
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

    // chunks of the contraction: n_ch of 64 elements, the last with vc valid 16-byte pieces (8 = full)
    const int d = lay.head_dim;
    const int n_ch = __builtin_amdgcn_readfirstlane((d + 63) >> 6);
    const int vc = __builtin_amdgcn_readfirstlane((d - 64 * (n_ch - 1)) >> 3);
    const bool last_partial = vc < 8;

    // operand reads: row l&15 of a 16-row tile, piece 4 ks + (l >> 4); the same offset serves K (A) and Q (B)
    const int f_rd = j * kCkRow + ck_swz(j, h);               // k-step 1: ^ 64
    unsigned char* qtile = kbuf + kCkQOff + wave * kCkQTile;

    // DMA sources.  K: 1 KiB block blk = 4 j2 + wave (10 blocks: rows 8 blk .. 8 blk + 7); lane -> row 8 blk + (lane >> 3), LDS
    // piece slot lane & 7 = source piece (lane & 7) ^ ((row >> 1) & 7).  Q: block i = rows 8 i .. 8 i + 7 of the wave's 32, same
    // rule.  *_full: every piece of the chunk is inside head_dim; *_last: the layer's partial last chunk, pieces past head_dim
    // re-fetch piece 0.  The chunk's 128 c bytes and the step's tensor are wave-uniform (scalar offset / resource descriptor).
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    unsigned kd_full[3], kd_last[3];
#pragma unroll
    for (int j2 = 0; j2 < 3; ++j2) {
        const int blk = 4 * j2 + wave;
        const int rowu = 8 * blk + (lane >> 3);
        const int row = min(rowu, kTok - 1);
        const int ch = (lane & 7) ^ ((rowu >> 1) & 7);
        kd_full[j2] = (unsigned)((row * (int)lay.k_st + ch * 8) * 2);
        kd_last[j2] = (unsigned)((row * (int)lay.k_st + (ch < vc ? ch * 8 : 0)) * 2);
    }
    unsigned qd_full[2], qd_last[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int ch = (lane & 7) ^ ((4 * par + (lane >> 4)) & 7);
        const int px = p0 + wave * 32 + (lane >> 3);
        const unsigned rowb = (unsigned)((q_off + (int64_t)min(px, lay.hw - 1) * lay.q_sp) * 2);
        qd_full[par] = rowb + (unsigned)ch * 16u;
        qd_last[par] = rowb + (ch < vc ? (unsigned)ch * 16u : 0u);
    }
    // Q block i sits 8 i pixel rows further: a wave-uniform byte step.  hw is a multiple of 8 (tap_chunk_supported), so a block is
    // inside the layer for all of its lanes or for none; a block outside re-reads block 0's rows (or, when the whole wave is
    // outside, the layer's last row: px is clamped) and its results are never stored.
    const int q_rows_in = __builtin_amdgcn_readfirstlane(lay.hw - (p0 + wave * 32));
    const unsigned q_step8 = (unsigned)__builtin_amdgcn_readfirstlane(8 * (int)lay.q_sp * 2);   // bytes per 8 pixel rows
    unsigned q_s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q_s[i] = 8 * i < q_rows_in ? (unsigned)i * q_step8 : 0u;
    const unsigned k_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(k_off * 2));

    auto tensor = [](const void* p) -> __amdgpu_buffer_rsrc_t {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, -1, 0x00020000);
    };
    // the fetches of sub-step (step s, chunk c) into K buffer `buf` and this wave's Q tile
    auto dma = [&](int s, int c, int buf) {
        const unsigned cb = (unsigned)c * 128u;
        const bool lastp = last_partial && c == n_ch - 1;     // wave-uniform
        const __amdgpu_buffer_rsrc_t kt = tensor(sptr[2 * s + 1]);
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
            const int blk = 4 * j2 + wave;                    // wave-uniform
            if (blk < 10)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(kt, (lds_ptr_t)(kbuf + buf * kCkKBuf + blk * 1024), 16,
                                                         lastp ? kd_last[j2] : kd_full[j2], k_base + cb, 0, 0);
        }
        const __amdgpu_buffer_rsrc_t qt = tensor(sptr[2 * s]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qt, (lds_ptr_t)(qtile + i * 1024), 16, lastp ? qd_last[i & 1] : qd_full[i & 1],
                                                     q_s[i] + cb, 0, 0);
    };

    // Sub-step protocol (u = running sub-step number, K buffer u & 1):
    //   wait for this wave's DMAs of sub-step u; barrier (every wave's part of the K tile has landed, every wave has left
    //   sub-step u - 1, whose K buffer is the one the next fetch overwrites); operand reads + MFMAs; then the DMAs of sub-step
    //   u + 1 (into the other K buffer and into this wave's own Q tile, whose reads the MFMAs have consumed).  After a layer's
    //   last chunk the softmax of the two pixel groups runs with those DMAs in flight.
    const floatx4 cmask = premask_tile4(h);
    dma(0, 0, 0);
    int buf = 0;
    for (int s = 0; s < n_steps; ++s) {
        floatx4 c0[5], c1[5];
        const int s_next = min(s + 1, n_steps - 1);           // branch-free: the last step re-fetches itself
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        chunk_mfma<IN, true>(kbuf + buf * kCkKBuf, qtile, f_rd, last_partial && n_ch == 1, vc, h, c0, c1, cmask);
        buf ^= 1;
        if (n_ch > 1) dma(s, 1, buf); else dma(s_next, 0, buf);
        for (int c = 1; c < n_ch; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            chunk_mfma<IN, false>(kbuf + buf * kCkKBuf, qtile, f_rd, last_partial && c == n_ch - 1, vc, h, c0, c1, cmask);
            buf ^= 1;
            if (c + 1 < n_ch) dma(s, c + 1, buf); else dma(s_next, 0, buf);
        }
        if constexpr (IN::kBf16) {
            softmax20_accumulate_bf16<ACC_T, true>(c0, lay, h, run0);
            softmax20_accumulate_bf16<ACC_T, true>(c1, lay, h, run1);
        } else {
            softmax20_accumulate<ACC_T, FAST_EXP, true>(c0, lay, h, run0);
            softmax20_accumulate<ACC_T, FAST_EXP, true>(c1, lay, h, run1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the last (redundant) fetch has landed before the staging tile reuses the space
    __syncthreads();                                          // all K reads done

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

// q_extent = elements from the tensor's first to past its last addressed Q element (batch * q_sb): byte offsets stay in 32 bits
bool tap_chunk_supported(int in_dtype, int head_dim, int hw, int64_t q_sp, int64_t k_st, int64_t q_sb, int64_t q_sh, int64_t k_sb,
                         int64_t k_sh, int64_t q_extent, const void* q, const void* k)
{
    if ((in_dtype != 0 && in_dtype != 2) || head_dim < 8 || head_dim > kCkMaxHeadDim || head_dim % 8 != 0 || hw % 8 != 0) return false;
    const int64_t s[] = {q_sp, k_st, q_sb, q_sh, k_sb, k_sh};
    for (int64_t v : s)
        if (v % 8 != 0 || v < 0) return false;
    if (k_st * 77 >= (int64_t)1 << 30 || q_sp * (int64_t)hw >= (int64_t)1 << 30 || q_extent >= (int64_t)1 << 30) return false;
    return ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15) == 0;
}

template <typename IN, typename ACC_T, bool FAST>
static hipError_t launch_chunk_k(const TapLaunch& L, hipStream_t stream, int grid, size_t* lds_out)
{
    const size_t lds = tap_chunk_lds_bytes<ACC_T>();
    *lds_out = lds;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tap_chunk_kernel<IN, ACC_T, FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((tap_chunk_kernel<IN, ACC_T, FAST>), dim3(grid), dim3(256), lds, stream, L);
    return hipGetLastError();
}

// interleave: the launch mixes head dims (see the block mapping at the top of the kernel).  in_dtype / acc_dtype: 0 = fp16, 1 = f32
// (sums only), 2 = bf16 (bf16 pipelines: bf16 or f32 sums, the one bf16 softmax flavour).
hipError_t launch_tap_chunk(const TapLaunch& L0, int in_dtype, int acc_dtype, int fast_exp, int interleave, hipStream_t stream, int* grid_out,
                            int* lds_out)
{
    TapLaunch L = L0;
    const int per = (L.total_wgs + 7) / 8;
    L.wgs_per_xcd = interleave ? -per : per;
    const int grid = per * 8;
    *grid_out = grid;
    size_t lds = 0;
    hipError_t e;
    if (in_dtype == 2) {
        if (acc_dtype == 2) e = launch_chunk_k<InBF16, bf16_t, true>(L, stream, grid, &lds);
        else if (acc_dtype == 1) e = launch_chunk_k<InBF16, float, true>(L, stream, grid, &lds);
        else return hipErrorInvalidValue;
    } else if (in_dtype != 0) {
        return hipErrorInvalidValue;
    } else if (acc_dtype == 0) {
        e = fast_exp ? launch_chunk_k<InF16, _Float16, true>(L, stream, grid, &lds) : launch_chunk_k<InF16, _Float16, false>(L, stream, grid, &lds);
    } else if (acc_dtype == 1) {
        e = fast_exp ? launch_chunk_k<InF16, float, true>(L, stream, grid, &lds) : launch_chunk_k<InF16, float, false>(L, stream, grid, &lds);
    } else {
        return hipErrorInvalidValue;
    }
    *lds_out = (int)lds;
    return e;
}

}  // namespace daam

