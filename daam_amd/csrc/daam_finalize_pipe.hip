// x2 (32 -> 64) finalize of fp16 planes, software-pipelined (round 3): compute_global_heat_map's per-key
// bicubic -> clamp(min=0) -> mean (reference daam/trace.py:112-126) for every 32 x 32 key of a selection.
//
// Same arithmetic as finalize_up32_mfma_kernel (daam_finalize.hip: both bicubic passes on v_mfma_f32_32x32x16_f16, T as an
// fp16 hi + lo pair, clamp + accumulate as ONE v_max_i32 on the result of an MFMA chain that starts from the running sums),
// rebuilt around what bounded that kernel -- 630 cycles per half plane at 4 waves per SIMD although its own instruction stream
// runs in 418, and in 358-380 when the stages of consecutive planes overlap (tools/gen_ubench_fin.py):
//   * TWO waves per SIMD, each running a hand-scheduled two-deep software pipeline (tools/gen_fin_pipe.py ->
//     daam_finalize_pipe_asm.inc, one asm statement): the pass-2 MFMAs of plane i, the clamps of plane i-1, the hi / lo split
//     of plane i+1 and the pass-1 MFMAs of plane i+2 are in flight together, every VALU instruction sits in a fixed gap behind
//     an MFMA that does not depend on it, results alternate between two register sets (even / odd planes; summed at the end);
//   * planes reach the MFMAs through an LDS ring of 8 planes per workgroup filled by LDS-DMA (global_load_lds_dwordx4: each of
//     the two waves fetches one 1 KiB half of every plane and both read all of it; counted vmcnt + one s_barrier per plane, no
//     staging registers): a plane crosses L2 -> LDS once (wave-private rings fetched every plane twice: 6.7 TB/s of traffic,
//     the bound of the first version), 7 planes in flight per workgroup;
//   * workgroup = 2 waves = the two 32-column halves (nt) of the output for ONE (token, key chunk); a wave walks ALL keys of its
//     chunk from a host-built pointer table padded with an all-zero plane to a common even length (no per-wave remainder code,
//     no key cap);
//   * the same-size (64 x 64) keys of the selection ride along: every wave adds its share of them to the same accumulators with an
//     identity MFMA + the same one-v_max clamp BEFORE entering the loop, under the latency of the ring's first planes -- SDXL-1024's
//     finalize is ONE class kernel (a second kernel on a second stream cost 15 us of event fork / join per call);
//   * ~215 VGPRs -> 2 waves per SIMD.
//
// Round 6: the same pipeline for the other two dtypes a context's sums can have (the reference's finalize takes whatever the running
// sums are, daam/trace.py:111-116) -- the generator writes one schedule per dtype:
//   * bf16 planes: pass 1 on v_mfma_f32_32x32x16_bf16 -- the plane is its A operand as it is; the tap matrix is NOT a bf16 matrix (three taps
//     clamped onto a border column add up to 283/256: nine significant bits), so the host splits W = W' + E into two bf16 matrices (E = 1/256
//     at [0][0] and [63][31]) and pass 1 has a third MFMA whose A operand -- plane columns 0..7 | 24..31 -- is selected from the two pieces
//     the lane holds anyway (4 v_cndmask): 11 MFMAs per plane.  T, pass 2, clamp and the folded same-size keys (identity MFMA in bf16) as above;
//   * f32 planes (accumulate='float32'): 4 KiB per plane, ring of 8; a lane reads its A pieces as 16 floats and splits them into an
//     fp16 hi + lo pair exactly like pass 2 splits T (2^-22 relative, the same error pass 2 already has): 4 pass-1 MFMAs and 32 more
//     VALU instructions per plane.  The 16-byte pieces of a 128-byte plane row sit XOR-swizzled in the ring (piece ^ ((row >> 1) & 7), done
//     on the DMA's per-lane SOURCE address): the four ds_read_b128 of an A operand are conflict-free.  The folded same-size keys take the
//     VALU (every lane loads the elements it owns in the C/D layout, acc += max(P, 0)).
#include "daam_types.h"
#include "../../include/daam_hip.h"       // DAAM_F16 / DAAM_F32 / DAAM_BF16

namespace daam {


typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kPipeRing = 16;              // planes per workgroup ring (the generated schedule is per depth: tools/gen_fin_pipe.py)
constexpr int kSameBatch = 8;           // same-size keys whose pieces are fetched together (32 loads in flight per lane)
constexpr int kSameBatchF32 = 4;        // ... of f32 planes (128 dword loads in flight per lane)
constexpr int kPipePlane = 32 * 32 * 2; // bytes

// max(a, b) for b >= 0 through the integer order of the bit patterns (daam_finalize.hip: fin_max_nonneg)
__device__ __forceinline__ float pipe_max_nonneg(float a, float b) {
    const int x = __float_as_int(a), y = __float_as_int(b);
    return __int_as_float(x > y ? x : y);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

// DT = plane dtype: DAAM_F16, DAAM_BF16, DAAM_F32 (daam_hip.h)
// (2 waves per SIMD: the register budget the allocator must respect -- arch VGPRs + AGPRs <= 256)
template <int DT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void finalize_up32_pipe_kernel(const FinPipeLaunch L)
{
    constexpr bool F32 = DT == DAAM_F32, BF16 = DT == DAAM_BF16;
    constexpr unsigned kPlane = F32 ? 2u * kPipePlane : (unsigned)kPipePlane;  // bytes of a 32 x 32 plane
    __shared__ __align__(16) unsigned char ring[kPipeRing * kPipePlane];         // 32 KiB (16 planes of 2 KiB / 8 of 4 KiB), shared by the workgroup's two waves

    if (L.nk_pad < 4 || (L.nk_pad & 1)) return;               // the pipeline's prologue / drain assume >= 4 planes, an even count (host-padded)
    const int lane = threadIdx.x & 63;
    const int nt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, g = lane >> 5;
    const int tok = blockIdx.x, chunk = blockIdx.y;

    // this chunk's plane pointers: nk_pad real or all-zero planes + kPipeRing + 1 more entries the ring prefetches past the end
    const unsigned long long* key_ptrs = L.key_ptrs + (size_t)chunk * L.ptr_stride;
    // LDS-DMA: this wave fetches half nt of every plane (lane: 16 bytes) into half nt of the ring slot; both waves read all of it
    // f32 planes: a DMA instruction of wave nt fills ring bytes [1024 nt, 1024 nt + 1024) of a slot = plane rows 8 nt .. 8 nt + 7 (a second
    // one, 2048 bytes further in both address spaces, rows 16 + 8 nt ..): lane -> row r = 8 nt + (lane >> 3), ring piece lane & 7, which holds
    // SOURCE piece (lane & 7) ^ ((r >> 1) & 7) of the row (the key is the same 16 rows further)
    const unsigned r32 = 8u * nt + ((unsigned)lane >> 3);
    const unsigned goff = (unsigned)tok * kPlane + (F32 ? r32 * 128u + ((((unsigned)lane & 7u) ^ ((r32 >> 1) & 7u)) << 4)
                                                        : (unsigned)nt * 1024u + (unsigned)lane * 16u);
    const unsigned ring_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)(&ring[0]));
    const unsigned ring_half = ring_base + (unsigned)nt * 1024u;
    [[maybe_unused]] const unsigned lds_rd = ring_base + (unsigned)n * 64u + (unsigned)g * 16u;    // A piece: row n, columns 8g.. (+32 bytes: 16 + 8g..)
    // f32: row n = 128 bytes, the lane's floats 8g .. 8g + 7 (k-step 0) and 16 + 8g .. (k-step 1) are piece columns 2g, 2g + 1, 4 + 2g, 5 + 2g
    [[maybe_unused]] uintx4 lds_rd4;
    {
        const unsigned key = ((unsigned)n >> 1) & 7u, row = ring_base + (unsigned)n * 128u;
        lds_rd4[0] = row + (((2u * g) ^ key) << 4);
        lds_rd4[1] = row + (((2u * g + 1u) ^ key) << 4);
        lds_rd4[2] = row + (((2u * g + 4u) ^ key) << 4);
        lds_rd4[3] = row + (((2u * g + 5u) ^ key) << 4);
    }
    if constexpr (F32) {
#include "daam_finalize_pipe_prefill_f32.inc"
    } else {
#include "daam_finalize_pipe_prefill_r16.inc"
    }

    // ---- the same-size (64 x 64) keys of the selection, under the latency of the ring's first planes ------------------
    // out += max(P, 0) for this wave's 32 columns of every row, straight into the accumulators of the pipeline (C/D layout:
    // lane (n, g) holds rows 32 mt + 8 b + 4 g + r of column 32 nt + n): the plane rows are the A operand (lane: row 32 mt + n,
    // 8 contiguous columns 32 nt + 16 ks + 8 g ..) of an MFMA against a 0 / 1 selection matrix -- products with 1.0 and sums with
    // zeros are exact, the chain starts from the running sums (D = acc + P), and acc + max(P, 0) == max(D, acc).
    // WHEN: workgroups alternate -- even ones before their x2 loop (under the latency of the ring's first planes), odd ones
    // after it -- so that at any time about half of a CU's waves stream same-size planes from HBM (no arithmetic to speak of)
    // while the other half run the issue-bound x2 loop with a SIMD to themselves (measured with every workgroup streaming first:
    // 15 us in which no x2 plane was computed, tools/exp/pipe_timing.py).
    floatx16 accA0 = {0}, accA1 = {0};                         // even planes x output row halves (mt); the pipeline adds to them
    // (workgroups are dispatched breadth-first, one per CU per sweep of 256: consecutive sweeps alternate, so every CU hosts both kinds)
    const bool same_first = (((blockIdx.y * gridDim.x + blockIdx.x) >> 8) & 1) == 0;
    auto same_size_keys = [&]() {
    if constexpr (F32) {
        // f32 planes: no operand conversion pays here -- every lane fetches the elements it owns in the C/D layout directly (one dword per
        // output row: lanes 0..31 / 32..63 of an instruction read two whole 128-byte lines), acc += max(P, 0) on the VALU; kSameBatchF32 keys
        // in flight per round trip
        if (L.same_per > 0) {
            const unsigned long long* sp = L.same_ptrs + (size_t)chunk * L.same_per;
            const unsigned soff = (unsigned)tok * (64u * 64u * 4u) + (unsigned)(4 * g) * 256u + (unsigned)(32 * nt + n) * 4u;
            for (int j0 = 0; j0 < L.same_per; j0 += kSameBatchF32) {
                float v[kSameBatchF32][32];
                unsigned long long ptr[kSameBatchF32];
#pragma unroll
                for (int j = 0; j < kSameBatchF32; ++j) ptr[j] = j0 + j < L.same_per ? sp[j0 + j] : 0ull;
#pragma unroll
                for (int j = 0; j < kSameBatchF32; ++j) {
                    const char* base = reinterpret_cast<const char*>(ptr[j] ? ptr[j] : ptr[0]);   // padding slot: re-reads the first key, result unused
                    if (base) {                                                  // wave-uniform
#pragma unroll
                        for (int i = 0; i < 32; ++i)
                            v[j][i] = *as_global<float>(base + soff + (unsigned)(32 * (i >> 4) + 8 * ((i & 15) >> 2) + (i & 3)) * 256u);
                    }
                }
#pragma unroll
                for (int j = 0; j < kSameBatchF32; ++j) {
                    if (!ptr[j]) continue;                                       // wave-uniform
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        accA0[i] += fmaxf(v[j][i], 0.f);
                        accA1[i] += fmaxf(v[j][16 + i], 0.f);
                    }
                }
            }
        }
    } else
    if (L.same_per > 0) {
        half8 sel[2];                                          // 1.0 in the plane dtype (bf16: 0x3f80), bit patterns in a half8
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned short one = BF16 ? 0x3f80 : 0x3c00, bits = (n == 16 * ks + 8 * g + e) ? one : (unsigned short)0;
                sel[ks][e] = __builtin_bit_cast(_Float16, bits);
            }
        auto ident = [&](half8 a, half8 b, floatx16 c) -> floatx16 {
            if constexpr (BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        };
        const unsigned long long* sp = L.same_ptrs + (size_t)chunk * L.same_per;
        const unsigned soff = (unsigned)tok * (64u * 64u * 2u) + (unsigned)n * 128u + (unsigned)(32 * nt + 8 * g) * 2u;
        // the pieces of kSameBatch keys are fetched together (one memory round trip per batch, not per key); padding entries are
        // null (wave-uniform)
        for (int j0 = 0; j0 < L.same_per; j0 += kSameBatch) {
            half8 a[kSameBatch][2][2];
            unsigned long long ptr[kSameBatch];
#pragma unroll
            for (int j = 0; j < kSameBatch; ++j) ptr[j] = j0 + j < L.same_per ? sp[j0 + j] : 0ull;
#pragma unroll
            for (int j = 0; j < kSameBatch; ++j) {
                // a padding slot re-reads the batch's first key (its result is not used); a chunk without any same-size key
                // (fewer keys than chunks) has ptr[0] == 0 and fetches nothing: no address is formed from a null pointer
                const char* base = reinterpret_cast<const char*>(ptr[j] ? ptr[j] : ptr[0]);
                if (base) {                                                      // wave-uniform
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            a[j][mt][ks] = *as_global<half8>(base + soff + mt * (32 * 128) + ks * 32);
                }
            }
#pragma unroll
            for (int j = 0; j < kSameBatch; ++j) {
                if (!ptr[j]) continue;                                           // wave-uniform
                floatx16 o0 = ident(a[j][0][0], sel[0], accA0);
                floatx16 o1 = ident(a[j][1][0], sel[0], accA1);
                o0 = ident(a[j][0][1], sel[1], o0);
                o1 = ident(a[j][1][1], sel[1], o1);
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    accA0[v] = pipe_max_nonneg(o0[v], accA0[v]);
                    accA1[v] = pipe_max_nonneg(o1[v], accA1[v]);
                }
            }
        }
    }
    };
    if (same_first) same_size_keys();
    // operand pieces of the banded tap matrix, built on the host (build_up32_ops in daam_api.hip):
    //   wx[ks][e]    = W[32nt + n][16ks + 8g + e]                              (B of pass 1)
    //   wy[t][ks][i] = W[32t + n][16ks + 8(i >> 2) + 4g + (i & 3)]             (A of pass 2, permuted k)
    //   bf16 planes: wx holds W' (bf16 bit patterns) and a seventh piece wxe[e] = E[32nt + n][g == 0 ? e : 24 + e], W = W' + E (see the generator)
    constexpr int kOps = BF16 ? 7 : 6;
    const DAAM_GLOBAL half8* ops = as_global<half8>(L.mfma_ops) + (size_t)(nt * 64 + lane) * kOps;
    // (fetched only now: 24 registers the batch of same-size pieces above needed; L2-resident, the same for every workgroup)
    const half8 wx0 = ops[0], wx1 = ops[1], wy00 = ops[2], wy01 = ops[3], wy10 = ops[4], wy11 = ops[5];
    [[maybe_unused]] const half8 wxe = ops[kOps - 1];

    int trips = __builtin_amdgcn_readfirstlane((L.nk_pad - 2) >> 1);              // steady-state loop trips, 2 planes each
    floatx16 accB0, accB1;                                                       // odd planes
    if constexpr (F32) {
#include "daam_finalize_pipe_asm_f32.inc"
    } else if constexpr (BF16) {
#include "daam_finalize_pipe_asm_bf16.inc"
    } else {
#include "daam_finalize_pipe_asm_r16.inc"
    }

    if (!same_first) same_size_keys();
    // C/D layout: lane (n, g) owns out[32 mt + 8 b + 4 g + r][32 nt + n] in register 4 b + r of tile mt
    float* out = L.out + (size_t)tok * 64 * 64 + 32 * nt + n;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int row = 32 * (i >> 4) + 8 * ((i & 15) >> 2) + 4 * g + (i & 3);
        const float v = (i < 16 ? accA0[i & 15] + accB0[i & 15] : accA1[i & 15] + accB1[i & 15]);
        atomicAdd(out + row * 64, v * L.inv_n);
    }
}

// acc_dtype: dtype of the planes (DAAM_F16 / DAAM_BF16 / DAAM_F32); L.mfma_ops must be the operand set of that dtype (bf16 planes: Wx in bf16)
hipError_t launch_finalize_up32_pipe(const FinPipeLaunch& L, int acc_dtype, hipStream_t stream, int* grid_out)
{
    dim3 grid(L.tokens, L.n_chunks);
    *grid_out = grid.x * grid.y;
    if (acc_dtype == DAAM_F32) hipLaunchKernelGGL(finalize_up32_pipe_kernel<DAAM_F32>, grid, dim3(128), 0, stream, L);
    else if (acc_dtype == DAAM_BF16) hipLaunchKernelGGL(finalize_up32_pipe_kernel<DAAM_BF16>, grid, dim3(128), 0, stream, L);
    else hipLaunchKernelGGL(finalize_up32_pipe_kernel<DAAM_F16>, grid, dim3(128), 0, stream, L);
    return hipGetLastError();
}

// planes the ring prefetches past a chunk's last one (pointer-table padding): 16 planes of 2 KiB, 8 of 4 KiB
int finalize_pipe_ring(int acc_dtype) { return acc_dtype == DAAM_F32 ? kPipeRing / 2 : kPipeRing; }

}  // namespace daam

