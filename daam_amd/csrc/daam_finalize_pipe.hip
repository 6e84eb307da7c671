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
#include "daam_types.h"

namespace daam {


typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kPipeRing = 16;              // planes per workgroup ring (the generated schedule is per depth: tools/gen_fin_pipe.py)
constexpr int kSameBatch = 8;           // same-size keys whose pieces are fetched together (32 loads in flight per lane)
constexpr int kPipePlane = 32 * 32 * 2; // bytes

// max(a, b) for b >= 0 through the integer order of the bit patterns (daam_finalize.hip: fin_max_nonneg)
__device__ __forceinline__ float pipe_max_nonneg(float a, float b) {
    const int x = __float_as_int(a), y = __float_as_int(b);
    return __int_as_float(x > y ? x : y);
}

// (2 waves per SIMD: the register budget the allocator must respect -- arch VGPRs + AGPRs <= 256)
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void finalize_up32_pipe_kernel(const FinPipeLaunch L)
{
    __shared__ __align__(16) unsigned char ring[kPipeRing * kPipePlane];         // 16 KiB, shared by the workgroup's two waves

    if (L.nk_pad < 4 || (L.nk_pad & 1)) return;               // the pipeline's prologue / drain assume >= 4 planes, an even count (host-padded)
    const int lane = threadIdx.x & 63;
    const int nt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, g = lane >> 5;
    const int tok = blockIdx.x, chunk = blockIdx.y;

    // this chunk's plane pointers: nk_pad real or all-zero planes + kPipeRing + 1 more entries the ring prefetches past the end
    const unsigned long long* key_ptrs = L.key_ptrs + (size_t)chunk * L.ptr_stride;
    // LDS-DMA: this wave fetches half nt of every plane (lane: 16 bytes) into half nt of the ring slot; both waves read all of it
    const unsigned goff = (unsigned)tok * kPipePlane + (unsigned)nt * 1024u + (unsigned)lane * 16u;
    const unsigned ring_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)(&ring[0]));
    const unsigned ring_half = ring_base + (unsigned)nt * 1024u;
    const unsigned lds_rd = ring_base + (unsigned)n * 64u + (unsigned)g * 16u;    // A piece: row n, columns 8g.. (+32 bytes: 16 + 8g..)
#include "daam_finalize_pipe_prefill_r16.inc"

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
    if (L.same_per > 0) {
        half8 sel[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) sel[ks][e] = (_Float16)((n == 16 * ks + 8 * g + e) ? 1.0f : 0.0f);
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
                floatx16 o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j][0][0], sel[0], accA0, 0, 0, 0);
                floatx16 o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j][1][0], sel[0], accA1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j][0][1], sel[1], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j][1][1], sel[1], o1, 0, 0, 0);
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
    const DAAM_GLOBAL half8* ops = as_global<half8>(L.mfma_ops) + (size_t)(nt * 64 + lane) * 6;
    // (fetched only now: 24 registers the batch of same-size pieces above needed; L2-resident, the same for every workgroup)
    const half8 wx0 = ops[0], wx1 = ops[1], wy00 = ops[2], wy01 = ops[3], wy10 = ops[4], wy11 = ops[5];

    int trips = __builtin_amdgcn_readfirstlane((L.nk_pad - 2) >> 1);              // steady-state loop trips, 2 planes each
    floatx16 accB0, accB1;                                                       // odd planes
#include "daam_finalize_pipe_asm_r16.inc"

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

hipError_t launch_finalize_up32_pipe(const FinPipeLaunch& L, hipStream_t stream, int* grid_out)
{
    dim3 grid(L.tokens, L.n_chunks);
    *grid_out = grid.x * grid.y;
    hipLaunchKernelGGL(finalize_up32_pipe_kernel, grid, dim3(128), 0, stream, L);
    return hipGetLastError();
}

int finalize_pipe_ring() { return kPipeRing; }

}  // namespace daam

