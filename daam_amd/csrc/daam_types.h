// Device-visible tables shared by the host API (daam_api.hip) and the kernels.
// gfx950 only; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace daam {

// Pointers that arrive through a device table are "generic" to the compiler, which then emits
// flat_load (counted on lgkmcnt as well as vmcnt, so every LDS wait also waits for HBM).
// Everything the tables hold is global memory: say so.
#define DAAM_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ const DAAM_GLOBAL T* as_global(const void* p) {
    return (const DAAM_GLOBAL T*)(p);
}
template <typename T> __device__ __forceinline__ DAAM_GLOBAL T* as_global_rw(void* p) {
    return (DAAM_GLOBAL T*)(p);
}

constexpr int kMaxTokens = 80;        // context_size is 77 (reference trace.py:194); padded tiles use 80 / 96
constexpr int kTapPixels = 64;        // query positions per workgroup tile (generic kernel)
constexpr int kMaxSlabSegs = 4;      // segments of a tap_slab_kernel launch (TapLaunch::seg_begin)
constexpr int kTapParts = 4;          // token chunks per pixel (one per wave of the 256-thread block)
constexpr int kTokPerPart = kMaxTokens / kTapParts;   // 20

// One layer's share of a tap launch.  All steps recorded for the layer in this launch have
// the same shape / strides (the host flushes otherwise).
struct TapLayer {
    void* acc;              // running sums [heads_kept, tokens, hw] (ctx acc dtype)
    int32_t heads_kept;     // BH - BH/2   (trace.py:240)
    int32_t bh_first;       // BH/2: first kept batch*heads index
    int32_t heads;          // H, to split bh -> (b, h)
    int32_t hw;
    int32_t head_dim;
    int32_t tiles_per_head; // ceil(hw / tile pixels)
    int32_t wg_begin;       // first (logical) workgroup of this layer in the launch
    int32_t n_steps;        // recorded steps of this layer in this launch
    int32_t ptr_begin;      // ptrs[ptr_begin + s] = step s
    int32_t round_logits;
    float scale;
    int32_t fresh;          // 1: running sums are known to be zero (first tap since reset): skip the read
    int64_t q_sb, q_sh, q_sp;
    int64_t k_sb, k_sh, k_st;
    // tap_slab_kernel: the entry covers the pixels [px_begin, px_end) of the layer in tiles of tile_px (a layer may have two entries: its
    // last pixels in half-size tiles at the end of the launch); every other kernel has [0, hw) and its own tile size
    int32_t px_begin, px_end, tile_px;
};

struct TapPtr {
    const void* q;
    const void* k;
};

// Kernel argument block.  `layers == nullptr` selects the by-value single-call form
// (immediate daam_tap_qk: no table upload, one layer, one step).
struct TapLaunch {
    const TapLayer* layers;
    const TapPtr* ptrs;
    int32_t n_layers;
    int32_t tokens;
    int32_t total_wgs;      // logical workgroups (grid is rounded up to a multiple of 8 XCDs)
    int32_t wgs_per_xcd;    // ceil(total_wgs / 8)
    unsigned* started;      // NULL, or a counter every (logical) workgroup bumps when it starts: the start gate of a multi-kernel
                            // flush (daam_tap_flush) holds the large kernel back until the small ones' workgroups are resident
    TapLayer one;
    TapPtr one_ptr;
    // tap_slab_kernel (daam_tap_slab.hip): the launch's workgroups are listed segment by segment -- [seg_begin[k], seg_begin[k + 1]) = table
    // entries of one cost (a head_dim's layers, or a pixel range of them) -- and every XCD takes an eighth of each segment, in order
    int32_t n_seg;
    int32_t seg_begin[kMaxSlabSegs + 1];
};

struct ProbsLaunch {        // daam_tap_probs
    void* acc;
    const void* probs;      // [BH, hw, tokens] contiguous
    int32_t heads_kept, bh_first, hw, tokens, tiles_per_head, total_wgs, wgs_per_xcd;
};

struct AttendLaunch {       // daam_attend: one cross-attention call, every (batch, head)
    const void* q;
    const void* k;
    const void* v;
    void* out;
    void* acc;              // the layer's running sums [heads_kept, tokens, hw], or NULL: no tap in this launch
    int32_t batch, heads, hw, head_dim, tiles_per_head, total_wgs, wgs_per_xcd;
    int32_t bh_first;       // first tapped batch*heads index (BH/2, trace.py:240)
    int32_t round_logits;
    int32_t fresh;          // sums are known to be zero: write instead of read-modify-write
    float scale;
    int64_t q_sb, q_sh, q_sp;
    int64_t k_sb, k_sh, k_st;
    int64_t v_sb, v_sh, v_st;
    int64_t o_sb, o_sh, o_sp;
};

constexpr int kFinMaxChunks = 31;

// One selected (layer, head) key of a finalize launch.
struct FinKey {
    const void* base;       // plane of token 0: [tokens, side, side] follows
    int32_t side;
    int32_t tab;            // bicubic table index (-1: side == out_side, identity)
};

struct FinLaunch {
    const FinKey* keys;
    const int16_t* tab_idx; // [n_tabs][out_side][4] border-clamped tap indices
    const float* tab_w;     // [n_tabs][out_side][4] weights (A = -0.75)
    float* out;             // [tokens, out_side, out_side]
    int32_t n_keys;
    int32_t n_chunks;
    int32_t tokens;
    int32_t out_side;
    float inv_n;
    int32_t max_side;       // largest non-identity side among the keys (LDS carve-up)
    const void* mfma_ops;   // x2 MFMA finalize: [2 nt][64 lanes][6] 16-byte operand pieces (host-built), or NULL
    // x2 MFMA finalize: chunk c covers the keys [chunk_begin[c], chunk_begin[c + 1]) (even boundaries; its two key lanes take
    // them alternately); see finalize_chunk_ranges() in daam_api.hip.
    int16_t chunk_begin[kFinMaxChunks + 1];
};

// x2 finalize, software-pipelined kernel (daam_finalize_pipe.hip): workgroup (token, chunk) walks the plane pointers
// key_ptrs[chunk * ptr_stride + 0 .. nk_pad) (token 0's plane of each key, or the all-zero plane as padding; nk_pad even, >= 4,
// the same for every chunk; the ring prefetches kPipeRing + 1 entries past nk_pad, which must be valid pointers too).
struct FinPipeLaunch {
    const unsigned long long* key_ptrs;
    const unsigned long long* same_ptrs;   // [n_chunks][same_per] planes of token 0 of the 64 x 64 keys folded in (0 = padding), or NULL
    int32_t same_per;
    const void* mfma_ops;   // as FinLaunch::mfma_ops
    float* out;             // [tokens, 64, 64]
    int32_t n_chunks;
    int32_t nk_pad;
    int32_t ptr_stride;     // entries per chunk in key_ptrs
    int32_t tokens;
    float inv_n;
};

// bfloat16 storage type (no arithmetic): values cross to f32 by a shift, back by round-to-nearest-even
struct bf16_t { uint16_t bits; };
__host__ __device__ inline float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = (uint32_t)v.bits << 16;
    return c.f;
}
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    bf16_t r;
    if ((c.u & 0x7fffffffu) > 0x7f800000u) { r.bits = (uint16_t)((c.u >> 16) | 0x0040u); return r; }   // NaN stays NaN
    r.bits = (uint16_t)((c.u + 0x7fffu + ((c.u >> 16) & 1u)) >> 16);
    return r;
}

}  // namespace daam
