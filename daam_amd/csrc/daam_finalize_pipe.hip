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
//   * planes reach the MFMAs through a wave-private LDS ring of 8 planes filled by LDS-DMA (global_load_lds_dwordx4, two 1 KiB
//     pieces per plane, counted vmcnt, no barrier, no staging registers): 7 planes = 14 KiB in flight per wave, 112 KiB per CU;
//   * workgroup = 2 waves = the two 32-column halves (nt) of the output for ONE (token, key chunk); a wave walks ALL keys of its
//     chunk from a host-built pointer table padded with an all-zero plane to a common even length (no per-wave remainder code,
//     no key cap);
//   * ~212 VGPRs -> 2 waves per SIMD; the remaining quarter of the register file takes the waves of the same-size class kernel,
//     which the host launches on a second stream (an HBM stream beside an issue-bound kernel).
#include "daam_types.h"

namespace daam {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kPipeRing = 8;            // planes per wave ring (must match R in tools/gen_fin_pipe.py)
constexpr int kPipePlane = 32 * 32 * 2; // bytes

__global__ __launch_bounds__(128) void finalize_up32_pipe_kernel(const FinPipeLaunch L)
{
    __shared__ __align__(16) unsigned char ring[2][kPipeRing * kPipePlane];      // 32 KiB: 4 workgroups per CU

    if (L.nk_pad < 4 || (L.nk_pad & 1)) return;               // the pipeline's prologue / drain assume >= 4 planes, an even count (host-padded)
    const int lane = threadIdx.x & 63;
    const int nt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, g = lane >> 5;
    const int tok = blockIdx.x, chunk = blockIdx.y;

    // operand pieces of the banded tap matrix, built on the host (build_up32_ops in daam_api.hip):
    //   wx[ks][e]    = W[32nt + n][16ks + 8g + e]                              (B of pass 1)
    //   wy[t][ks][i] = W[32t + n][16ks + 8(i >> 2) + 4g + (i & 3)]             (A of pass 2, permuted k)
    const DAAM_GLOBAL half8* ops = as_global<half8>(L.mfma_ops) + (size_t)(nt * 64 + lane) * 6;
    const half8 wx0 = ops[0], wx1 = ops[1], wy00 = ops[2], wy01 = ops[3], wy10 = ops[4], wy11 = ops[5];

    // this chunk's plane pointers: nk_pad real or all-zero planes + kPipeRing + 1 more entries the ring prefetches past the end
    const unsigned long long* key_ptrs = L.key_ptrs + (size_t)chunk * L.ptr_stride;
    const unsigned goff_lo = (unsigned)tok * kPipePlane + (unsigned)lane * 16u;   // lane's 16 bytes of a plane's first KiB
    const unsigned goff_hi = goff_lo + 1024u;
    const unsigned ring_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)(&ring[nt][0]));
    const unsigned lds_rd = ring_base + (unsigned)n * 64u + (unsigned)g * 16u;    // A piece: row n, columns 8g.. (+32 bytes: 16 + 8g..)
    int trips = __builtin_amdgcn_readfirstlane((L.nk_pad - 2) >> 1);              // steady-state loop trips, 2 planes each

    floatx16 accA0, accA1, accB0, accB1;                                         // even / odd planes x output row halves (mt)
#include "daam_finalize_pipe_asm.inc"

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
