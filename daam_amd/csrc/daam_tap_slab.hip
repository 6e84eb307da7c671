// Deferred fp16 tap for the SD-v1.x head dims (40 / 80 / 160) that reads Q and K in WHOLE 128-byte lines: one workgroup takes a
// 640-byte SLAB of the rows -- 8 / 4 / 2 adjacent heads -- for a tile of 32 pixels.  gfx950, v_mfma_f32_16x16x32_f16.
//
// Why (SD-v1.5, BASELINE.json configs[1]; round-4 verdict): tap_chunk_kernel's workgroup is (layer, head, 128 pixels) and fetches its
// head's 80- / 160- / 320-byte pieces of the 640- / 1280- / 2560-byte pixel rows.  The memory system moves 128-byte lines, the
// eight heads of a pixel tile run far apart in time, and the launch pulled 2.89 GB through the L2s for 1.27 GB of algorithmic bytes
// (profiles/r04_counters.json) -- at 6.5 TB/s of real traffic it was bound by bytes it did not need.  640 bytes is the smallest span
// that is whole lines AND whole heads for all three head dims (lcm(2 d, 128)), so here every line of Q is fetched once, by exactly
// one workgroup, and consumed completely: the slab's heads sit side by side in ONE LDS image [rows][640 B], each wave contracts
// its own head's columns of it.
//
// Work split: 8 waves.  head_dim 40: wave w = head w of the slab, 32 pixels as two groups of 16 (20 MFMAs + 2 softmaxes per step, as
// a wave of the other tap kernels).  head_dim 80: wave = (head w & 3, 16-pixel group w >> 2): 15 MFMAs + 1 softmax.  head_dim 160:
// waves 0..3 = (head w & 1, group w >> 1): 25 MFMAs + 1 softmax; waves 4..7 only help with the fetches.  Same operand layout, k
// order (k-step ks of a head = its elements 32 ks .. 32 ks + 31, lane quarter h the eight from 8 h), MFMA chain and softmax
// (daam_tap16_softmax.h) as tap_chunk_kernel / tap_d64_kernel / tap_wide_kernel: the sums it leaves are bit-identical to theirs
// (tests/test_gpu_slab.py).
//
// LDS (71.0 KB, two workgroups per CU = 4 waves per SIMD): K slab [80 rows][640 B] (77 tokens; rows 77..79 finite filler whose
// logits are masked), Q tile [32 pixel rows][640 B], 16 zero bytes, the per-step tensor pointers.  A row is 40 sixteen-byte
// pieces; piece p of row r sits at slot p ^ ((r >> 1) & 7) (an XOR inside aligned groups of 8 pieces = inside one 128-byte line):
// the operand reads (16 rows x one piece per ds_read_b128 lane quarter) are conflict-free for head_dim 80 / 160 and take 4.5
// instead of 4 LDS cycles on average for head_dim 40 (tools/exp/slab_banks.py checks the gfx950 lane groups).  Both operands arrive
// by LDS-DMA (buffer_load_dwordx4 ... lds): the LDS image of a wave-instruction is lane-linear, the swizzle is applied to the
// SOURCE piece, and a wave-instruction covers 8 whole lines.  The k-step that holds the tail of a head (head_dim 40: piece 4 of
// 5; 80: pieces 8, 9 of 10) reads its missing Q pieces from the 16 zero bytes and the matching K pieces from a valid piece of
// the same head (finite x 0 = 0).
//
// Step protocol (both operands single-buffered; raw s_barrier, counted waits -- a __syncthreads() would drain the DMAs):
//   vmcnt(0), barrier          K(s) and Q(s) are in LDS
//   Q operands -> registers; lgkmcnt(0), barrier; DMA Q(s + 1)       (in flight for the rest of the step)
//   K operands + MFMAs; barrier; DMA K(s + 1)                        (an L2 hit; lands during the softmax)
//   softmax + accumulate (running sums stay in registers for the whole launch)
#include "daam_tap16_softmax.h"

// Cache policy of the Q fetches: non-temporal (2).  Every Q line is read exactly once per launch, by one workgroup; with nt the once-read
// lines no longer push the K slabs, which every tile of a layer re-reads, out of the L2s: SD-v1.5 2517 -> 2583 maps/s, tap 0.351 -> 0.342 ms,
// alternating three times on one box (LABNOTES R5.9; neutral on the head_dim-64 kernel, whose K tiles are 5x smaller per byte of Q).
constexpr int kSlabQAux = 2;

namespace daam {

constexpr int kSlabBytes = 640;                          // bytes of a Q / K row a workgroup takes: five 128-byte lines
constexpr int kSlabSlots = kSlabBytes / 16;              // 40 sixteen-byte pieces per row
constexpr int kSlabPx = 32;                              // pixels per workgroup
constexpr int kSlabWaves = 8;
constexpr int kSlabKBytes = kD64Rows * kSlabBytes;       // 51200: 80 rows
constexpr int kSlabQOff = kSlabKBytes;
constexpr int kSlabQBytes = kSlabPx * kSlabBytes;        // 20480
constexpr int kSlabZeroOff = kSlabQOff + kSlabQBytes;    // 71680: 16 zero bytes
constexpr int kSlabPtrOff = kSlabZeroOff + 16;           // per-step tensor pointers
constexpr int kSlabKInstr = (kTok * kSlabSlots + 63) / 64;   // 49 wave-instructions cover rows 0..76 (the last one runs into row 78)
constexpr int kSlabQInstr = kSlabPx * kSlabSlots / 64;       // 20

template <typename ACC_T> constexpr size_t tap_slab_lds_bytes() {
    const size_t loop = (size_t)kSlabPtrOff + (size_t)kMaxStepsPerLaunch * 2 * sizeof(void*);
    const size_t st = (size_t)kSlabWaves * kTok * kSlabPx * sizeof(ACC_T);       // wave-private staging of the sums (f32: 78848)
    return loop > st ? loop : st;
}

// blockIdx -> logical workgroup.  The launch's workgroups are listed segment by segment (a segment = the layers of one head_dim, all
// of one cost); XCD x (= blockIdx & 7) takes the x-th eighth of EVERY segment, in segment order: every XCD gets the same mix of light
// and heavy workgroups, and the tiles that share a K slab (consecutive logical workgroups) meet in one or two L2s.
__device__ __forceinline__ int slab_logical_block(const TapLaunch& L)
{
    const int x = blockIdx.x & 7;
    int i = blockIdx.x >> 3;
    for (int k = 0; k < L.n_seg; ++k) {
        const int n = L.seg_begin[k + 1] - L.seg_begin[k];
        const int lo = (int)(((long long)n * x) >> 3), hi = (int)(((long long)n * (x + 1)) >> 3);
        if (i < hi - lo) return L.seg_begin[k] + lo + i;
        i -= hi - lo;
    }
    return -1;
}

// TP = pixels per workgroup: 32, or -- head_dim 40 only -- 16: the half-size workgroups that take the last pixels of the head_dim-40
// layers at the end of the launch (one 16-pixel group per wave, half the step chain's work: the chip drains more evenly)
template <int D, int TP, typename ACC_T, bool FAST_EXP>
__device__ __forceinline__ void slab_body(unsigned char* smem, const TapLaunch& L, const TapLayer& lay, int wg)
{
    constexpr int PPH = D / 8;                                // 16-byte pieces per head row: 5 / 10 / 20
    constexpr int NH = kSlabSlots / PPH;                      // heads per slab: 8 / 4 / 2
    constexpr int NKS = (PPH + 3) / 4;                        // k-steps of 32 elements: 2 / 3 / 5 (the last one partial for 40 and 80)
    constexpr int G = NH == 8 ? TP / 16 : 1;                  // 16-pixel groups per wave
    constexpr int ITEMS = NH * (TP / 16 / G);                 // waves with arithmetic to do: 8 / 8 / 4
    constexpr int TW = 16 * G;                                // pixels per wave
    static_assert(TP == kSlabPx || (TP == 16 && D == 40), "half-size tiles: head_dim 40 only");
    constexpr int VEC = AccVec<ACC_T>::kPerVec;
    constexpr int PPRW = TW / VEC;                            // 16-byte pieces per staged row of a wave
    static_assert(NH * PPH == kSlabSlots && (PPH % 4 == 0 || NKS * 4 - PPH < 4), "slab geometry");
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // scalar: DMA blocks and roles must not become exec masks
    const int j = lane & 15, h = lane >> 4;
    const bool active = ITEMS == kSlabWaves || wave < ITEMS;  // wave-uniform (head_dim 160: waves 4..7 only fetch)
    const int head = wave % NH, grp = G == 2 ? 0 : wave / NH; // this wave's head of the slab and (G == 1) its 16-pixel group

    const int n_steps = lay.n_steps;
    const int rel = wg - lay.wg_begin;
    const int slab = rel / lay.tiles_per_head;                // tiles_per_head = tiles per slab here
    const int p0 = lay.px_begin + (rel - slab * lay.tiles_per_head) * TP;
    const int kh0 = slab * NH;                                // first kept head of the slab
    const int bh = lay.bh_first + kh0;
    const int b = bh / lay.heads, hd0 = bh - b * lay.heads;
    const int64_t k_off = b * lay.k_sb + hd0 * lay.k_sh;      // k_sh == q_sh == D (tap_slab_supported): the slab's heads are adjacent
    const int64_t q_off = b * lay.q_sb + hd0 * lay.q_sh;

    // ---- running sums -> registers, through a wave-private staging tile [77][TW] -------------------------------------------
    typename Pair<ACC_T>::T run0[kSlots16 / 2], run1[kSlots16 / 2];   // slot pairs (2i, 2i+1); run1 = the second group (G == 2)
    ACC_T* stage = reinterpret_cast<ACC_T*>(smem) + (size_t)wave * kTok * TW;
    ACC_T* acc = reinterpret_cast<ACC_T*>(lay.acc) + (size_t)(kh0 + head) * kTok * lay.hw;
    const int px0 = p0 + 16 * grp;                            // first pixel of this wave
    if (!lay.fresh && active) {
        for (int piece = lane; piece < kTok * PPRW; piece += 64) {
            const int row = piece / PPRW, col = (piece - row * PPRW) * VEC;
            if (px0 + col < lay.px_end)
                *reinterpret_cast<float4v*>(stage + row * TW + col) = *as_global<float4v>(acc + (size_t)row * lay.hw + px0 + col);
        }
#pragma unroll
        for (int i = 0; i < kSlots16; ++i) {
            const int t = slot16_token(i, h);
            if (t < kTok) {
                run0[i >> 1][i & 1] = from_acc<ACC_T>(stage[t * TW + j]);
                if constexpr (G == 2) run1[i >> 1][i & 1] = from_acc<ACC_T>(stage[t * TW + 16 + j]);
            } else {
                run0[i >> 1][i & 1] = 0;
                if constexpr (G == 2) run1[i >> 1][i & 1] = 0;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < kSlots16; ++i) { run0[i >> 1][i & 1] = 0; run1[i >> 1][i & 1] = 0; }
    }
    __syncthreads();                                          // the staging tiles (f32: they reach into the pointer table) are free

    // ---- one-time LDS contents: step pointers, the zero piece, K rows 77..79 (the DMAs rewrite row 77 and part of 78 with row 76) ----
    const void** sptr = reinterpret_cast<const void**>(smem + kSlabPtrOff);
    {
        const DAAM_GLOBAL TapPtr* ptrs = as_global<TapPtr>(L.ptrs) + lay.ptr_begin;
        for (int i = tid; i < n_steps; i += 64 * kSlabWaves) {
            sptr[2 * i] = ptrs[i].q;
            sptr[2 * i + 1] = ptrs[i].k;
        }
        if (tid < 4) reinterpret_cast<unsigned*>(smem + kSlabZeroOff)[tid] = 0u;
        for (int i = tid; i < 3 * kSlabSlots; i += 64 * kSlabWaves)
            *reinterpret_cast<float4v*>(smem + kTok * kSlabBytes + i * 16) = float4v{0, 0, 0, 0};
    }
    __syncthreads();

    // ---- DMA sources.  Wave-instruction i of an image covers its slots 64 i .. 64 i + 63 (LDS bytes 1024 i ..): slot sigma = row
    // sigma / 40, position t = sigma % 40, which holds source piece t ^ ((row >> 1) & 7) of that row.  Ten instructions are exactly 16
    // rows, so instructions i and i + 10 m share ONE per-lane byte offset (same t, same swizzle key) and differ by a wave-uniform
    // 16 m rows:
    //   K  i = w + 10 m, m = 0..4, and i = 8 + (w & 1) + 10 (w >> 1)                     (i = 0..47: token rows 0..76.8)
    //      waves 4..7 also i = 48 (the rest of row 76; its lanes past row 76 re-read row 76 -- four waves write the same bytes)
    //   Q  i = w + 10 m, m = 0, 1                                                        (i = 0..7, 10..17)
    //      waves 0..3 also i = 8 + (w & 1) + 10 (w >> 1)                                  (8, 9, 18, 19)
    // Every wave issues nine instructions per step; the only branches are the two wave-uniform ones around the extras.
    // hw is a multiple of 16 (tap_slab_supported): pixel rows 0..15 of the tile are inside the layer, rows 16..31 all or none (then they
    // re-read rows 0..15; their results are never stored).
    auto lane_off = [&](int i, int64_t row_stride, int row0, int row_max) -> unsigned {      // byte offset of this lane's piece in instruction i
        const int sigma = 64 * i + lane;
        const int row = sigma / kSlabSlots, t = sigma - row * kSlabSlots;
        return (unsigned)((row0 + min(row, row_max)) * (int)row_stride * 2) + (unsigned)((t ^ ((row >> 1) & 7)) << 4);
    };
    const int wx = 8 + (wave & 1);                            // base instruction of this wave's instruction from the classes 8, 9
    const unsigned kdA = lane_off(wave, lay.k_st, 0, kTok - 1), kdB = lane_off(wx, lay.k_st, 0, kTok - 1);
    const unsigned qdA = lane_off(wave, lay.q_sp, p0, 15);
    const unsigned xd = wave < 4 ? lane_off(wx, lay.q_sp, p0, 15) : lane_off(kSlabKInstr - 1, lay.k_st, 0, kTok - 1);
    const unsigned k_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(k_off * 2));
    const unsigned q_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(q_off * 2));
    const unsigned k16 = (unsigned)__builtin_amdgcn_readfirstlane(16 * (int)lay.k_st * 2);                        // bytes per 16 token rows
    const unsigned q16 = (unsigned)__builtin_amdgcn_readfirstlane(lay.px_end - p0 >= kSlabPx ? 16 * (int)lay.q_sp * 2 : 0);   // ... 16 pixel rows
    const unsigned kB_s = k_base + (unsigned)(wave >> 1) * k16;
    const unsigned xq_s = q_base + (unsigned)(wave >> 1) * q16;
    const int kB_lds = (wx + 10 * (wave >> 1)) * 1024;
    const int xq_lds = kSlabQOff + (wx + 10 * (wave >> 1)) * 1024;
    auto tensor = [](const void* p) -> __amdgpu_buffer_rsrc_t {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, -1, 0x00020000);
    };
    auto dma_k = [&](int s) {                                 // six instructions (waves 4..7: seven)
        const __amdgpu_buffer_rsrc_t kt = tensor(sptr[2 * s + 1]);
#pragma unroll
        for (int m = 0; m < 5; ++m)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(kt, (lds_ptr_t)(smem + (wave + 10 * m) * 1024), 16, kdA, k_base + (unsigned)m * k16, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(kt, (lds_ptr_t)(smem + kB_lds), 16, kdB, kB_s, 0, 0);
        if (wave >= 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(kt, (lds_ptr_t)(smem + (kSlabKInstr - 1) * 1024), 16, xd, k_base, 0, 0);
    };
    auto dma_q = [&](int s) {                                 // two instructions (waves 0..3: three)
        const __amdgpu_buffer_rsrc_t qt = tensor(sptr[2 * s]);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(qt, (lds_ptr_t)(smem + kSlabQOff + wave * 1024), 16, qdA, q_base, 0, kSlabQAux);
        if constexpr (TP == kSlabPx)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qt, (lds_ptr_t)(smem + kSlabQOff + (wave + 10) * 1024), 16, qdA, q_base + q16, 0, kSlabQAux);
        if (wave < TP / 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(qt, (lds_ptr_t)(smem + xq_lds), 16, xd, xq_s, 0, kSlabQAux);   // 16-pixel tiles: instructions 8, 9 only
    };

    // ---- operand reads: lane (j, h) of k-step ks takes piece 4 ks + h of its head from row j of a 16-row tile; the same offset
    // serves K (A: token rows 16 mt + j) and Q (B: pixel rows 16 g + j) because the swizzle key depends on (row >> 1) & 7 only
    unsigned f_k[NKS], f_q[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int pi = 4 * ks + h;
        const bool valid = pi < PPH;
        // (lanes past the head's last piece in the tail k-step: K from a valid piece of the same head x the ZERO piece of Q = 0 for finite K.
        // The chunked / wide kernels zero-pad K instead; the two differ only when that K piece holds inf / NaN (0 x inf = NaN here) -- inputs for
        // which the reference's softmax yields NaN rows for the head as well)
        const int p = head * PPH + (valid ? pi : 4 * ks);
        f_k[ks] = (unsigned)(j * kSlabBytes + ((p ^ ((j >> 1) & 7)) << 4));
        f_q[ks] = valid ? (unsigned)(kSlabQOff + 16 * grp * kSlabBytes) + f_k[ks] : (unsigned)kSlabZeroOff;
    }
    // second pixel group of a head_dim-40 wave: 16 rows further, except for the lanes that read the zero piece
    [[maybe_unused]] const unsigned f_q1_last = (4 * (NKS - 1) + h) < PPH ? f_q[NKS - 1] + 16u * kSlabBytes : (unsigned)kSlabZeroOff;

    const floatx4 cmask = premask_tile4(h);
    dma_k(0);
    dma_q(0);
    for (int s = 0; s < n_steps; ++s) {
        const int s_next = min(s + 1, n_steps - 1);           // branch-free: the last step re-fetches itself
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");          // every wave's part of K(s) and Q(s) has landed
        half8 qv[G][NKS];
        if (active) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                qv[0][ks] = *reinterpret_cast<const half8*>(smem + f_q[ks]);
                if constexpr (G == 2)
                    qv[1][ks] = *reinterpret_cast<const half8*>(smem + (ks == NKS - 1 ? f_q1_last : f_q[ks] + 16u * kSlabBytes));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // the Q tile has been read by everyone
        dma_q(s_next);
        floatx4 c[G][5];
        if (active) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
                    const half8 a = *reinterpret_cast<const half8*>(smem + mt * 16 * kSlabBytes + f_k[ks]);
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        c[g][mt] = InF16::mfma(a, qv[g][ks], ks == 0 ? (mt == 4 ? cmask : floatx4{0, 0, 0, 0}) : c[g][mt]);   // tokens 77..79: -inf from the start
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // the K slab has been read by everyone
        dma_k(s_next);
        if (active) {
            softmax20_accumulate<ACC_T, FAST_EXP, true>(c[0], lay, h, run0);
            if constexpr (G == 2) softmax20_accumulate<ACC_T, FAST_EXP, true>(c[G - 1], lay, h, run1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");              // nothing of the last (redundant) fetches is in flight: LDS is free

    // ---- write back: registers -> wave-private [token][pixel] tile -> 16-byte row pieces -------------------------------------
    if (active) {
#pragma unroll
        for (int i = 0; i < kSlots16; ++i) {
            const int t = slot16_token(i, h);
            if (t < kTok) {
                stage[t * TW + j] = to_acc<ACC_T>(run0[i >> 1][i & 1]);
                if constexpr (G == 2) stage[t * TW + 16 + j] = to_acc<ACC_T>(run1[i >> 1][i & 1]);
            }
        }
        for (int piece = lane; piece < kTok * PPRW; piece += 64) {
            const int row = piece / PPRW, col = (piece - row * PPRW) * VEC;
            if (px0 + col < lay.px_end)
                *as_global_rw<float4v>(acc + (size_t)row * lay.hw + px0 + col) = *reinterpret_cast<const float4v*>(stage + row * TW + col);
        }
    }
}

template <typename ACC_T, bool FAST_EXP>
__global__ __launch_bounds__(64 * kSlabWaves, 4) void tap_slab_kernel(const TapLaunch L)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int wg = slab_logical_block(L);
    if (wg < 0) return;
    tap_mark_started(L);
    TapLayer lay;
    const DAAM_GLOBAL TapLayer* gl = as_global<TapLayer>(L.layers);
    load_layer(gl + mfma_find_layer(gl, L.n_layers, wg), &lay);
    switch (lay.head_dim) {                                   // wave-uniform
    case 40:
        if (lay.tile_px == 16) slab_body<40, 16, ACC_T, FAST_EXP>(smem, L, lay, wg);
        else slab_body<40, kSlabPx, ACC_T, FAST_EXP>(smem, L, lay, wg);
        break;
    case 80: slab_body<80, kSlabPx, ACC_T, FAST_EXP>(smem, L, lay, wg); break;
    default: slab_body<160, kSlabPx, ACC_T, FAST_EXP>(smem, L, lay, wg); break;
    }
}

// heads per 640-byte slab for a head_dim the kernel takes, else 0
int tap_slab_heads(int head_dim) { return head_dim == 40 ? 8 : head_dim == 80 ? 4 : head_dim == 160 ? 2 : 0; }
int tap_slab_tile_pixels() { return kSlabPx; }

// q_extent = elements from the tensor's first to past its last addressed Q element (batch * q_sb): byte offsets stay in 32 bits
bool tap_slab_supported(int in_dtype, int batch, int heads, int head_dim, int hw, int64_t q_sp, int64_t k_st, int64_t q_sb, int64_t q_sh,
                        int64_t k_sb, int64_t k_sh, int64_t q_extent, const void* q, const void* k)
{
    const int nh = tap_slab_heads(head_dim);
    if (in_dtype != 0 || !nh || hw % 16 != 0) return false;   // hw % 16: the second half of a pixel tile is inside the layer or outside, never split
    // the slab's heads are adjacent columns of one row, and the kept heads (batch * heads / 2 onwards, trace.py:240) start on a slab
    if (q_sh != head_dim || k_sh != head_dim || heads % nh != 0 || ((batch * heads) / 2) % nh != 0) return false;
    const int64_t s[] = {q_sp, k_st, q_sb, k_sb};
    for (int64_t v : s)
        if (v % 8 != 0 || v < 0) return false;
    if (q_sp < (int64_t)heads * head_dim || k_st < (int64_t)heads * head_dim) return false;
    if (k_st * 77 >= (int64_t)1 << 30 || q_sp * (int64_t)hw >= (int64_t)1 << 30 || q_extent >= (int64_t)1 << 30) return false;
    return ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15) == 0;
}

template <typename ACC_T, bool FAST>
static hipError_t launch_slab_k(const TapLaunch& L, hipStream_t stream, int grid, size_t* lds_out)
{
    const size_t lds = tap_slab_lds_bytes<ACC_T>();
    *lds_out = lds;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tap_slab_kernel<ACC_T, FAST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((tap_slab_kernel<ACC_T, FAST>), dim3(grid), dim3(64 * kSlabWaves), lds, stream, L);
    return hipGetLastError();
}

// L.seg_begin / L.n_seg describe the segments (see slab_logical_block); acc_dtype 0 = fp16, 1 = f32 sums
hipError_t launch_tap_slab(const TapLaunch& L, int acc_dtype, int fast_exp, hipStream_t stream, int* grid_out, int* lds_out)
{
    if (L.n_seg < 1 || L.n_seg > kMaxSlabSegs || !L.layers) return hipErrorInvalidValue;
    int per = 0;                                              // workgroups of the fullest XCD
    for (int x = 0; x < 8; ++x) {
        int n_x = 0;
        for (int k = 0; k < L.n_seg; ++k) {
            const long long n = L.seg_begin[k + 1] - L.seg_begin[k];
            n_x += (int)((n * (x + 1)) >> 3) - (int)((n * x) >> 3);
        }
        per = n_x > per ? n_x : per;
    }
    const int grid = per * 8;
    *grid_out = grid;
    size_t lds = 0;
    hipError_t e;
    if (acc_dtype == 0) e = fast_exp ? launch_slab_k<_Float16, true>(L, stream, grid, &lds) : launch_slab_k<_Float16, false>(L, stream, grid, &lds);
    else if (acc_dtype == 1) e = fast_exp ? launch_slab_k<float, true>(L, stream, grid, &lds) : launch_slab_k<float, false>(L, stream, grid, &lds);
    else return hipErrorInvalidValue;
    *lds_out = (int)lds;
    return e;
}

}  // namespace daam

