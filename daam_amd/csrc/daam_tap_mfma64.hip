// fp16 tap for head_dim = 64 (SDXL: every layer; SD-2.x), gfx950: both MFMA operands travel
// HBM/L2 -> LDS by DMA (global_load_lds_dwordx4: no VGPR round trip, no operand prefetch registers),
// so the step loop holds only the MFMA accumulators / softmax state / running sums in registers
// (4 waves per SIMD) and every Q row is fetched as one fully coalesced 128-byte line.
//
// Same arithmetic and rounding points as daam_tap_mfma.hip (shared softmax_accumulate()).
//
// LDS (37 KiB per workgroup of 256 threads = 4 waves = 128 pixels of one (layer, kept head)):
//   kbuf[2] : K of step s / s+1, [80 rows][8 pieces of 16 B]          2 x 10 KiB
//   qbuf    : Q of the 4 waves, each [32 pixel rows][8 pieces]            16 KiB (wave-private 4 KiB)
//   sptr    : per-step (q, k) base pointers                                1 KiB
//   (the write-back staging tile [77][128] aliases kbuf + qbuf)
// Piece p of row r lives at r*128 + ((p ^ (r & 7) ^ ((r >> 3) & 1)) * 16): with that XOR swizzle the
// ds_read_b128 operand fetches (lane = row, fixed piece) are bank-conflict free and the DMA writes
// stay lane-linear (lane l of DMA instruction i lands at base + i*1024 + l*16, so it simply FETCHES
// the piece that belongs there: row 8i + l/8, piece (l&7) ^ (row&7) ^ ((row>>3)&1) - the 8 lanes of a
// row still cover one 128-byte line).
// Step loop, one barrier per step:
//   top barrier (every wave has waited for its own K / Q DMA of this step)
//   B = 4 x ds_read_b128 (own Q rows), A = 12 x ds_read_b128 (K), 12 MFMA
//   issue the DMA of step s+1 (K -> other buffer, Q -> own region: its reads are done)
//   softmax + accumulate (registers)           <- the DMA flies under this
//   s_waitcnt vmcnt(0)
#include "daam_tap_common.h"

namespace daam {

constexpr int kRowB = 128;                         // bytes per 64-element fp16 row
constexpr int kKRows = 80;                         // 77 token rows, rounded to whole DMA instructions
constexpr int kKBuf = kKRows * kRowB;              // 10240
constexpr int kQWave = 32 * kRowB;                 // 4096
constexpr int kOperandBytes = 2 * kKBuf + 4 * kQWave;   // 36864

template <typename ACC_T> constexpr size_t tap_mfma64_lds_bytes() {
    const size_t st = (size_t)kTok * kMfmaPixels * sizeof(ACC_T);
    return (st > (size_t)kOperandBytes ? st : (size_t)kOperandBytes) + (size_t)kMaxStepsPerLaunch * 2 * sizeof(void*);
}

__device__ __forceinline__ int swz(int row, int piece) { return piece ^ (row & 7) ^ ((row >> 3) & 1); }

__device__ __forceinline__ void dma16(const _Float16* src, unsigned char* lds_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const DAAM_GLOBAL void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_uniform, 16, 0, 0);
}

template <typename ACC_T, bool FAST_EXP>
__global__ __launch_bounds__(256, (sizeof(ACC_T) == 2 ? 4 : 2)) void tap_mfma64_kernel(const TapLaunch L)
{
    constexpr int VEC = AccVec<ACC_T>::kPerVec;
    constexpr int PPR = kMfmaPixels / VEC;
    constexpr size_t kPtrOff = tap_mfma64_lds_bytes<ACC_T>() - (size_t)kMaxStepsPerLaunch * 2 * sizeof(void*);

    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* kbuf = smem;                               // [2][kKBuf]
    unsigned char* qbuf = smem + 2 * kKBuf;                   // [4][kQWave]
    ACC_T* stage = reinterpret_cast<ACC_T*>(smem);            // [kTok][kMfmaPixels], aliases the operands
    const void** sptr = reinterpret_cast<const void**>(smem + kPtrOff);

    const int wg = mfma_logical_block(L.total_wgs, L.wgs_per_xcd);
    if (wg < 0) return;
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
    const int b = bh / lay.heads, h = bh - b * lay.heads;
    const int64_t k_off = b * lay.k_sb + h * lay.k_sh;
    const int64_t q_off = b * lay.q_sb + h * lay.q_sh;

    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, g = lane >> 5;

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

    // ---- DMA geometry (fixed for the launch) ------------------------------------------------------
    // lane l of a DMA instruction serves row (l >> 3) of the instruction's 8-row block, LDS slot l & 7
    const int sub = lane >> 3, slot = lane & 7;
    // K: wave w issues instruction blocks i = w, w+4 (and w+8 for w < 2): rows 8i + sub, clamped to 76
    int k_src[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int row = 8 * (wave + 4 * j) + sub;
        k_src[j] = min(row, kTok - 1) * (int)lay.k_st + swz(row, slot) * 8;         // elements
    }
    // Q: wave-local rows 8i + sub, i = 0..3 -> pixel p0 + 32*wave + row (clamped inside the map)
    int64_t q_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + sub;
        const int pixel = min(p0 + wave * 32 + row, lay.hw - 1);
        q_src[i] = q_off + (int64_t)pixel * lay.q_sp + swz(row, slot) * 8;
    }
    auto dma_step = [&](int s, int buf) {
        const _Float16* kp = reinterpret_cast<const _Float16*>(sptr[2 * s + 1]) + k_off;
        const _Float16* qp = reinterpret_cast<const _Float16*>(sptr[2 * s]);
        unsigned char* kdst = kbuf + buf * kKBuf + wave * 1024;
        dma16(kp + k_src[0], kdst);
        dma16(kp + k_src[1], kdst + 4 * 1024);
        if (wave < 2) dma16(kp + k_src[2], kdst + 8 * 1024);
        unsigned char* qdst = qbuf + wave * kQWave;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(qp + q_src[i], qdst + i * 1024);
    };
    // operand read addresses: A rows mt*32 + n, B row n (wave-private), pieces 2ks + g
    const unsigned char* b_rd = qbuf + wave * kQWave + n * kRowB;

    dma_step(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int s = 0; s < n_steps; ++s) {
        __syncthreads();
        const unsigned char* kb = kbuf + (s & 1) * kKBuf;
        half8 bq[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bq[ks] = *reinterpret_cast<const half8*>(b_rd + swz(n, 2 * ks + g) * 16);
        floatx16 c0 = {0}, c1 = {0}, c2 = {0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int p = 2 * ks + g;
            const half8 a0 = *reinterpret_cast<const half8*>(kb + (n) * kRowB + swz(n, p) * 16);
            const half8 a1 = *reinterpret_cast<const half8*>(kb + (32 + n) * kRowB + swz(32 + n, p) * 16);
            const half8 a2 = *reinterpret_cast<const half8*>(kb + (64 + n) * kRowB + swz(64 + n, p) * 16);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bq[ks], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bq[ks], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, bq[ks], c2, 0, 0, 0);
        }
        // every LDS read of this step has been consumed by an MFMA: the Q region (wave-private) and the
        // other K buffer (not read since the previous barrier) may be overwritten by the next step's DMA
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (s + 1 < n_steps) dma_step(s + 1, (s + 1) & 1);
        softmax_accumulate<ACC_T, FAST_EXP>(c0, c1, c2, lay, g, run);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                          // all operand reads done before the staging tile reuses the space

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

// head_dim 64, rows 128-byte aligned: the DMA kernel applies
bool tap_mfma64_supported(int head_dim, int64_t q_sp, int64_t k_st, int64_t q_sb, int64_t q_sh, int64_t k_sb,
                          int64_t k_sh, const void* q, const void* k)
{
    if (head_dim != 64) return false;
    const int64_t s[] = {q_sp, k_st, q_sb, q_sh, k_sb, k_sh};
    for (int64_t v : s)
        if (v % 8 != 0) return false;
    // global_load_lds element offsets are kept in 32 bits for K
    if (k_st * 77 >= (int64_t)1 << 30) return false;
    return ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15) == 0;
}

template <typename ACC_T, bool FAST>
static hipError_t launch64(const TapLaunch& L, hipStream_t stream, int grid, size_t* lds_out)
{
    const size_t lds = tap_mfma64_lds_bytes<ACC_T>();
    *lds_out = lds;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tap_mfma64_kernel<ACC_T, FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((tap_mfma64_kernel<ACC_T, FAST>), dim3(grid), dim3(256), lds, stream, L);
    return hipGetLastError();
}

hipError_t launch_tap_mfma64(const TapLaunch& L, int acc_dtype, int fast_exp, hipStream_t stream, int* grid_out,
                             int* lds_out)
{
    const int grid = L.wgs_per_xcd * 8;
    *grid_out = grid;
    size_t lds = 0;
    hipError_t e;
    if (fast_exp) e = acc_dtype == 0 ? launch64<_Float16, true>(L, stream, grid, &lds) : launch64<float, true>(L, stream, grid, &lds);
    else e = acc_dtype == 0 ? launch64<_Float16, false>(L, stream, grid, &lds) : launch64<float, false>(L, stream, grid, &lds);
    *lds_out = (int)lds;
    return e;
}

}  // namespace daam
