// fp16 tap for head_dim <= 64 (64: every SDXL / SD-2.x layer; 40: the 64x64 layers of SD-v1.5, zero-padded
// to 64), gfx950, built on v_mfma_f32_16x16x32_f16.
//
// Why a second tiling: with 32x32 MFMA tiles a lane owns one pixel and 40 token slots, so the
// softmax state (logits, exponentials, running sums) plus three 16-register accumulator tiles push
// the kernel to 160+ VGPRs = 3 waves per SIMD, and the phase timers show the kernel is latency /
// occupancy bound there (VALU ~50 % busy).  With 16x16 tiles the 77 tokens of a pixel are spread
// over FOUR lanes (20 slots each), the live softmax state per 16-pixel group halves, and a wave
// processes its 32 pixels as two groups whose MFMA and softmax phases interleave: 4 waves per SIMD.
//
// Same arithmetic / rounding points as daam_tap_mfma.hip:
//   logits = fp16(f32(q.k) * scale) -> f32 softmax -> fp16(p) -> acc += p (accumulator dtype).
//
// Workgroup = 4 waves = 128 pixels of one (layer, kept head) -- or, for head_dim-64 launches with fp16 sums since round 4, 8 waves = 256
// pixels sharing ONE K tile (template parameter WAVES); wave w: pixels
// [32w, 32w+32) = groups 0 / 1 of 16.  "Swapped" product S^T = K Q^T: A = K rows (lane: token row
// l&15 of the 16-row tile, k = 8*(l>>4)..+7 of the 32-wide k-step), B = Q^T (lane: pixel l&15, same
// k split).  C/D: lane holds pixel l&15 and tokens 16*mt + 4*(l>>4) + r (mt = 0..4, r = 0..3) = 20 slots.
// Both operands reach the MFMAs through LDS in FULL 128-byte rows: K of a step as [80 rows][128 B],
// double-buffered; Q of a step as a wave-private [32 pixel rows][128 B] tile.  Either is fetched from
// HBM one step ahead by coalesced loads (eight consecutive lanes = one whole row) -- straight into LDS by LDS-DMA for
// head_dim 64 (round 3), through staging registers otherwise -- with its 16-byte chunks XOR-swizzled by (row >> 1) & 7,
// which makes the operand reads
// (16 rows x one chunk per ds_read_b128) conflict-free without padding.  Round 2 fetched Q straight into
// the MFMA layout (16 rows x 64 B per load instruction): the same bytes, but 16 bytes per L1 tag
// lookup -- 0.67 lookups per cycle per CU and a read-tag-conflict stall in 20 % of the cycles
// (TCP_TOTAL_CACHE_ACCESSES, TCP_READ_TAGCONFLICT_STALL_CYCLES; profiles/r02_tap_tcp.txt).
#include "daam_tap16_softmax.h"

// Data path of the head_dim-64 launches (FULL64: every SDXL / SD-2.x layer), as measured over rounds 1-5 (LABNOTES): K and Q go HBM -> LDS by
// LDS-DMA (buffer_load ... lds: the swizzle is applied to the SOURCE address, the LDS image of a wave-instruction is lane-linear) instead of
// through staging registers + ds_write_b128 (24 VGPRs fewer per lane-step); the next step's DMAs go out AHEAD of this step's MFMAs (K right
// after the barrier, Q once the wave's four operand reads have returned); EIGHT waves per workgroup share ONE K tile (256 pixels of a head:
// half the K traffic per pixel; two workgroups per CU = the same 4 waves per SIMD, 53 KB of LDS each).  head_dim < 64 (zero-padded chunks
// cannot come from a DMA) takes the register-staged form on four waves.  The forms that were measured and dropped -- register-staged FULL64,
// DMAs behind the MFMAs, four-wave FULL64, a second Q buffer two steps ahead, head-minor numbering, TLB touches, non-temporal Q, the timing
// ablations -- live as patches under tools/exp/patches/ (tools/exp/build_variant.sh --lab), not in this file.

namespace daam {

constexpr int kTapRow = 128;                        // bytes per K / Q row in LDS (head_dim 64 x fp16), chunks swizzled
constexpr int kTapKBuf = kD64Rows * kTapRow;        // 10240: 80 K rows, rows 77..79 stay zero
constexpr int kTapQTile = 32 * kTapRow;             // 4096: one wave's 32 pixel rows
constexpr int kTapQOff = 2 * kTapKBuf;              // Q tiles of the four waves follow the two K buffers

// WAVES = waves per workgroup: 4 (128 pixels) or 8 (256 pixels of one head, ONE K tile for twice the pixels)
template <typename ACC_T, int WAVES = 4> constexpr size_t tap_d64_lds_bytes() {
    const size_t kb = 2 * (size_t)kTapKBuf + WAVES * (size_t)kTapQTile, st = (size_t)kTok * (32 * WAVES) * sizeof(ACC_T);
    return (kb > st ? kb : st) + (size_t)kMaxStepsPerLaunch * 2 * sizeof(void*);     // fp16 sums: 37888 -> 4 workgroups per CU
}

// byte offset of 16-byte chunk `chunk` inside row `row` of a swizzled [rows][128 B] image
__device__ __forceinline__ constexpr int swz_chunk(int row, int chunk) { return ((chunk ^ ((row >> 1) & 7)) << 4); }

// FULL64: every layer of the launch has head_dim == 64 (SDXL): the zero-padding selects of the head_dim < 64 case (8 VALU per
// wave-step) are compiled out
template <typename IN, typename ACC_T, bool FAST_EXP, bool FULL64, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES, ((WAVES == 8 || (sizeof(ACC_T) == 2 && !IN::kBf16)) ? 4 : 3)) void tap_d64_kernel(const TapLaunch L)
{
    constexpr int NT = 64 * WAVES;                            // threads per workgroup
    constexpr int TILE = 32 * WAVES;                          // pixels per workgroup (the host sizes tiles_per_head with it)
    static_assert(WAVES == 4 || (WAVES == 8 && FULL64), "eight waves: head_dim-64 launches (the DMA path) only");
    constexpr int KCH = (kTok * 8 + 255) / 256;               // 16-B K pieces per thread per step (3)
    constexpr int VEC = AccVec<ACC_T>::kPerVec;
    constexpr int PPR = TILE / VEC;
    constexpr size_t kPtrOff = tap_d64_lds_bytes<ACC_T, WAVES>() - (size_t)kMaxStepsPerLaunch * 2 * sizeof(void*);

    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* kbuf = smem;                               // [2][kTapKBuf], then the four waves' Q tiles
    ACC_T* stage = reinterpret_cast<ACC_T*>(smem);            // [kTok][TILE], aliases both
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
        for (int i = tid; i < lay.n_steps; i += NT) {
            sptr[2 * i] = ptrs[i].q;
            sptr[2 * i + 1] = ptrs[i].k;
        }
    } else if (tid == 0) {
        sptr[0] = L.one_ptr.q;
        sptr[1] = L.one_ptr.k;
    }
    const int n_steps = lay.n_steps;
    const int rel = wg - lay.wg_begin;
    const int kh = rel / lay.tiles_per_head;                  // (head, tile) numbering: the tiles of a head share its K tile out of one L2
    const int p0 = (rel - kh * lay.tiles_per_head) * TILE;
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
        for (int piece = tid; piece < kTok * PPR; piece += NT) {
            const int row = piece / PPR, col = (piece - row * PPR) * VEC;
            if (p0 + col < lay.hw)
                *reinterpret_cast<float4v*>(stage + row * TILE + col) =
                    *as_global<float4v>(acc + (size_t)row * lay.hw + p0 + col);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kSlots16; ++i) {
            const int t = slot16_token(i, h);
            if (t < kTok) {
                run0[i >> 1][i & 1] = from_acc<ACC_T>(stage[t * TILE + wave * 32 + j]);
                run1[i >> 1][i & 1] = from_acc<ACC_T>(stage[t * TILE + wave * 32 + 16 + j]);
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
    // head_dim < 64 (multiple of 8; SD-v1.5's 40): the contraction runs over 64 with zeros beyond head_dim --
    // K chunks past it are never written (the buffers are zeroed once), Q chunks past it are fetched from a valid
    // address and cleared before they are written to LDS.
    const int d = lay.head_dim;
    const bool partial = !FULL64 && d < 64;                   // wave-uniform
    if (partial) {
        for (int i = tid; i < 2 * kTapKBuf / 16; i += NT)
            *reinterpret_cast<float4v*>(kbuf + i * 16) = float4v{0, 0, 0, 0};
        __syncthreads();                                      // the first K tile lands on top of the zeros
    } else {
        // K rows 77..79 (never written by a step) must be finite: zero them once, both buffers
        for (int i = tid; i < 2 * 3 * (kTapRow / 16); i += NT) {
            const int buf = i / (3 * (kTapRow / 16)), r = i % (3 * (kTapRow / 16));
            *reinterpret_cast<float4v*>(kbuf + buf * kTapKBuf + kTok * kTapRow + r * 16) = float4v{0, 0, 0, 0};
        }
    }

    // per-thread K piece coordinates: piece c = tid + 256 j2 -> row t = c / 8 = (tid >> 3) + 32 j2, chunk c % 8.  The swizzle
    // key ((t >> 1) & 7) and the validity of the chunk do not depend on j2, so ONE LDS offset (+ 4096 j2) and ONE global
    // offset (+ 32 rows for piece 1, folded into the scalar base; piece 2 has its own, because the threads whose row would
    // be 77..95 re-read their piece 0 instead) serve the three pieces.
    static_assert(KCH == 3, "three K pieces per thread");
    const int k_t = tid >> 3, k_ch = tid & 7;
    const bool k_in_row = k_ch * 8 < d;                       // chunk inside the head's d elements
    const bool k_row2 = k_t + 64 < kTok;                      // piece 2 exists
    const unsigned k_src0 = (unsigned)((k_t * (int)lay.k_st + (k_in_row ? k_ch * 8 : 0)) * 2);
    const int k_dst0 = k_t * kTapRow + swz_chunk(k_t, k_ch);
    const unsigned k_step = (unsigned)__builtin_amdgcn_readfirstlane(32 * (int)lay.k_st * 2);   // bytes per 32 K rows
    const unsigned k_src2 = k_row2 ? k_src0 + 2 * k_step : k_src0;
    // Q pieces of this lane: piece p = lane + 64 i (i = 0..3) of the wave's 32 pixel rows -> row = p >> 3 = (lane >> 3) + 8 i,
    // chunk = lane & 7.  Addresses = wave-uniform base (the step's tensor pointer, made an SGPR pair by readfirstlane; + 8 i
    // pixel rows for piece i) + ONE 32-bit per-lane byte offset that never changes.  tap_d64_supported() keeps every
    // offset below 2^31.
    const int q_row = lane >> 3, q_chunk = lane & 7;
    const bool q_valid = q_chunk * 8 < d;
    const unsigned q_co = q_valid ? q_chunk * 16 : 0;
    const int q_px = p0 + wave * 32 + q_row;
    const unsigned q_b0 = (unsigned)((q_off + (int64_t)min(q_px, lay.hw - 1) * lay.q_sp) * 2) + q_co;
    // piece i sits 8 i pixel rows further: a wave-uniform byte step added to the scalar base.  hw is a multiple of 8
    // (tap_d64_supported), so a piece is inside the layer for all of its lanes or for none; a piece outside re-reads piece 0's
    // rows (or, when the whole wave is outside, the layer's last row: q_b0 is clamped) and its results are never stored.
    const int q_rows_in = __builtin_amdgcn_readfirstlane(lay.hw - (p0 + wave * 32));
    const unsigned q_step8 = (unsigned)__builtin_amdgcn_readfirstlane(8 * (int)lay.q_sp * 2);   // bytes per 8 pixel rows
    unsigned q_s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q_s[i] = 8 * i < q_rows_in ? (unsigned)i * q_step8 : 0u;
    // LDS write position of piece i: row (q_row + 8 i), swizzle key ((lane >> 4) + 4 i) & 7 = (lane >> 4) ^ 4 (i & 1)
    unsigned char* qtile = kbuf + kTapQOff + wave * kTapQTile;
    const int q_wr = q_row * kTapRow + ((q_chunk ^ (lane >> 4)) << 4);
    // operand reads: row l&15 of a 16-row tile, chunk 4 ks + (l >> 4); the same offset serves K (A) and Q (B)
    const int f_rd = j * kTapRow + swz_chunk(j, h);            // k-step 1: ^ 64
    // Fetches are raw buffer loads: address = the step's tensor (a wave-uniform resource descriptor built from the pointer
    // in SGPRs) + a per-lane 32-bit byte offset that never changes + a wave-uniform byte offset.  No 64-bit address
    // arithmetic on the VALU (7 v_lshl_add_u64 per wave-step with plain global loads), offsets stay single registers.
    auto tensor = [](const void* p) -> __amdgpu_buffer_rsrc_t {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, -1, 0x00020000);
    };
    const unsigned k_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(k_off * 2));

    float4v kreg[KCH];
    auto issue_k = [&](int s) {
        const __amdgpu_buffer_rsrc_t kt = tensor(sptr[2 * s + 1]);
        kreg[0] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(kt, k_src0, k_base, 0));
        kreg[1] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(kt, k_src0, k_base + k_step, 0));
        kreg[2] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(kt, k_src2, k_base, 0));
    };
    auto commit_k = [&](int buf) {
        unsigned char* kb = kbuf + buf * kTapKBuf + k_dst0;
        if (k_in_row) {
            *reinterpret_cast<float4v*>(kb) = kreg[0];
            *reinterpret_cast<float4v*>(kb + 32 * kTapRow) = kreg[1];
            if (k_row2) *reinterpret_cast<float4v*>(kb + 64 * kTapRow) = kreg[2];
        }
    };
    float4v qreg[4];
    auto issue_q = [&](int s) {
        const __amdgpu_buffer_rsrc_t qt = tensor(sptr[2 * s]);
#pragma unroll
        for (int i = 0; i < 4; ++i) qreg[i] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(qt, q_b0, q_s[i], 0));
    };
    auto commit_q = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4v v = (partial && !q_valid) ? float4v{0, 0, 0, 0} : qreg[i];
            *reinterpret_cast<float4v*>(qtile + i * 8 * kTapRow + (q_wr ^ (64 * (i & 1)))) = v;
        }
    };

    // LDS-DMA form (FULL64 launches only).  K: 1 KiB block blk = WAVES j2 + wave (10 blocks: rows 8 blk .. 8 blk + 7; rows 77..79 re-read
    // row 76: finite, their logits are masked); lane -> row 8 blk + (lane >> 3), LDS chunk slot lane & 7 = source chunk
    // (lane & 7) ^ ((row >> 1) & 7).  Q: block i = rows 8 i .. 8 i + 7 of the wave's 32, same rule.
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    unsigned kd_src[3];
#pragma unroll
    for (int j2 = 0; j2 < 3; ++j2) {
        const int blk = WAVES * j2 + wave;
        const int row = min(8 * blk + (lane >> 3), kTok - 1);
        const int ch = (lane & 7) ^ (((8 * blk + (lane >> 3)) >> 1) & 7);
        kd_src[j2] = (unsigned)((row * (int)lay.k_st + ch * 8) * 2);
    }
    unsigned qd_src[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int ch = (lane & 7) ^ ((4 * par + (lane >> 4)) & 7);
        const int px = p0 + wave * 32 + (lane >> 3);
        qd_src[par] = (unsigned)((q_off + (int64_t)min(px, lay.hw - 1) * lay.q_sp) * 2) + (unsigned)ch * 16u;
    }
    auto dma_k = [&](int s, int buf) {
        const __amdgpu_buffer_rsrc_t kt = tensor(sptr[2 * s + 1]);
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
            const int blk = WAVES * j2 + wave;                // wave-uniform
            if (blk < 10)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(kt, (lds_ptr_t)(kbuf + buf * kTapKBuf + blk * 1024), 16, kd_src[j2], k_base, 0, 0);
        }
    };
    auto dma_q = [&](int s) {
        const __amdgpu_buffer_rsrc_t qt = tensor(sptr[2 * s]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qt, (lds_ptr_t)(qtile + i * 1024), 16, qd_src[i & 1], q_s[i], 0, 0);
    };
    const floatx4 cmask = premask_tile4(h);
    // one denoising step: logits of step s from the K and Q tiles in LDS, then the fetches of the next step (head_dim 64: by DMA
    // into the other K buffer / this wave's own Q tile, whose reads are behind it; head_dim < 64: step s + 1 from the staging
    // registers into LDS and the request for step s + 2), softmax + accumulate of the two pixel groups
    auto step = [&](int s) {
        __syncthreads();
        const unsigned char* kb = kbuf + (s & 1) * kTapKBuf;
        [[maybe_unused]] const int s_fetch = min(s + 1, n_steps - 1);   // branch-free: the last step re-fetches itself
        // the K buffer of step s + 1 was last read in step s - 1 and every wave is past this step's barrier: its DMAs go out first
        if constexpr (FULL64) dma_k(s_fetch, (s + 1) & 1);
        const half8 q00 = *reinterpret_cast<const half8*>(qtile + f_rd), q01 = *reinterpret_cast<const half8*>(qtile + (f_rd ^ 64));
        const half8 q10 = *reinterpret_cast<const half8*>(qtile + 16 * kTapRow + f_rd);
        const half8 q11 = *reinterpret_cast<const half8*>(qtile + 16 * kTapRow + (f_rd ^ 64));
        if constexpr (FULL64) {
            // this wave's Q tile is free once its four operand reads have returned: the next step's rows are requested BEFORE the MFMAs
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_q(s_fetch);
        }
        floatx4 c0[5], c1[5];
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            const half8 a0 = *reinterpret_cast<const half8*>(kb + mt * 16 * kTapRow + f_rd);
            const half8 a1 = *reinterpret_cast<const half8*>(kb + mt * 16 * kTapRow + (f_rd ^ 64));
            c0[mt] = IN::mfma(a0, q00, mt == 4 ? cmask : floatx4{0, 0, 0, 0});     // tokens 77..79: -inf from the start of their chain
            c1[mt] = IN::mfma(a0, q10, mt == 4 ? cmask : floatx4{0, 0, 0, 0});
            c0[mt] = IN::mfma(a1, q01, c0[mt]);
            c1[mt] = IN::mfma(a1, q11, c1[mt]);
        }
        if constexpr (!FULL64) {
            // head_dim < 64 (register-staged): the pieces of step s + 1 were requested a whole step ago -- into LDS now (K buffer
            // (s + 1) & 1 was last read in step s - 1, which every wave left before this step's barrier; the Q tile is this wave's
            // own and its operand reads are behind it), then the request for step s + 2 goes out: a fetch has a whole step to
            // land instead of one softmax (SD-v1.5's 64 x 64 layers: 330 -> 285 us per 50-step launch)
            commit_k((s + 1) & 1);
            commit_q();
            issue_k(min(s + 2, n_steps - 1));                 // branch-free: the last steps re-fetch the last one
            issue_q(min(s + 2, n_steps - 1));
        }
        if constexpr (IN::kBf16) {
            softmax20_accumulate_bf16<ACC_T, true>(c0, lay, h, run0);
            softmax20_accumulate_bf16<ACC_T, true>(c1, lay, h, run1);
        } else {
            softmax20_accumulate<ACC_T, FAST_EXP, true>(c0, lay, h, run0);
            softmax20_accumulate<ACC_T, FAST_EXP, true>(c1, lay, h, run1);
        }
        // this wave's DMAs have landed; the next step's barrier publishes K
        if constexpr (FULL64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if constexpr (FULL64) {
        dma_k(0, 0);
        dma_q(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        issue_k(0);
        issue_q(0);
        commit_k(0);
        commit_q();
        issue_k(min(1, n_steps - 1));
        issue_q(min(1, n_steps - 1));
    }
    for (int s = 0; s < n_steps; ++s) step(s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing of the last (redundant) fetches is in flight any more
    __syncthreads();                                          // all K reads done before the staging tile reuses the space

    // ---- write back: registers -> LDS [token][pixel] -> 16-byte row pieces -------------------
#pragma unroll
    for (int i = 0; i < kSlots16; ++i) {
        const int t = slot16_token(i, h);
        if (t < kTok) {
            stage[t * TILE + wave * 32 + j] = to_acc<ACC_T>(run0[i >> 1][i & 1]);
            stage[t * TILE + wave * 32 + 16 + j] = to_acc<ACC_T>(run1[i >> 1][i & 1]);
        }
    }
    __syncthreads();
    for (int piece = tid; piece < kTok * PPR; piece += NT) {
        const int row = piece / PPR, col = (piece - row * PPR) * VEC;
        if (p0 + col < lay.hw)
            *as_global_rw<float4v>(acc + (size_t)row * lay.hw + p0 + col) =
                *reinterpret_cast<const float4v*>(stage + row * TILE + col);
    }
}

bool tap_d64_supported(int head_dim, int hw, int64_t q_sp, int64_t k_st, int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_sh,
                       const void* q, const void* k)
{
    if (head_dim < 8 || head_dim > 64 || head_dim % 8 != 0 || hw % 8 != 0) return false;
    const int64_t s[] = {q_sp, k_st, q_sb, q_sh, k_sb, k_sh};
    for (int64_t v : s)
        if (v % 8 != 0) return false;
    if (k_st * 77 >= (int64_t)1 << 30) return false;         // K element offsets are kept in 32 bits
    return ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15) == 0;
}

template <typename IN, typename ACC_T, bool FAST, bool FULL64, int WAVES = 4>
static hipError_t launch_d64_k(const TapLaunch& L, hipStream_t stream, int grid, size_t lds)
{
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tap_d64_kernel<IN, ACC_T, FAST, FULL64, WAVES>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((tap_d64_kernel<IN, ACC_T, FAST, FULL64, WAVES>), dim3(grid), dim3(64 * WAVES), lds, stream, L);
    return hipGetLastError();
}

template <typename IN, typename ACC_T, bool FAST>
static hipError_t launch_d64(const TapLaunch& L, hipStream_t stream, int grid, size_t* lds_out, bool full64, int waves8)
{
    if (waves8 && full64) {                                   // every dtype pair fits 128 VGPRs (101-128), no spills
        const size_t lds8 = tap_d64_lds_bytes<ACC_T, 8>();
        *lds_out = lds8;
        return launch_d64_k<IN, ACC_T, FAST, true, 8>(L, stream, grid, lds8);
    }
    if (waves8) return hipErrorInvalidValue;                  // the host sized the tiles for eight waves: no other form may run them
    const size_t lds = tap_d64_lds_bytes<ACC_T>();
    *lds_out = lds;
    return full64 ? launch_d64_k<IN, ACC_T, FAST, true>(L, stream, grid, lds) : launch_d64_k<IN, ACC_T, FAST, false>(L, stream, grid, lds);
}

// tile pixels the host must size a head_dim-64 launch for: 256 when the eight-wave form takes it (every layer head_dim 64; fp16 Q / K
// with fp16 or f32 sums, bf16 Q / K with bf16 or f32 sums), else 128
int tap_d64_tile_pixels(int in_dtype, int acc_dtype, int full64)
{
    const bool pair = (in_dtype == 0 && (acc_dtype == 0 || acc_dtype == 1)) || (in_dtype == 2 && (acc_dtype == 2 || acc_dtype == 1));
    return (pair && full64) ? 256 : 128;
}

hipError_t launch_tap_d64(const TapLaunch& L, int in_dtype, int acc_dtype, int fast_exp, int full64, int waves8, hipStream_t stream, int* grid_out, int* lds_out)
{
    const int grid = L.wgs_per_xcd * 8;
    *grid_out = grid;
    size_t lds = 0;
    hipError_t e;
    if (in_dtype == 2) {                                       // bf16 pipeline: one softmax flavour
        if (acc_dtype == 2) e = launch_d64<InBF16, bf16_t, true>(L, stream, grid, &lds, full64 != 0, waves8);
        else if (acc_dtype == 1) e = launch_d64<InBF16, float, true>(L, stream, grid, &lds, full64 != 0, waves8);
        else return hipErrorInvalidValue;
    } else if (acc_dtype != 0 && acc_dtype != 1) {
        return hipErrorInvalidValue;
    } else if (fast_exp) {
        e = acc_dtype == 0 ? launch_d64<InF16, _Float16, true>(L, stream, grid, &lds, full64 != 0, waves8) : launch_d64<InF16, float, true>(L, stream, grid, &lds, full64 != 0, waves8);
    } else {
        e = acc_dtype == 0 ? launch_d64<InF16, _Float16, false>(L, stream, grid, &lds, full64 != 0, waves8) : launch_d64<InF16, float, false>(L, stream, grid, &lds, full64 != 0, waves8);
    }
    *lds_out = (int)lds;
    return e;
}

}  // namespace daam
