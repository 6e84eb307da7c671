// Host side of libdaam_hip.so: the C ABI declared in include/daam_hip.h.
// Owns no activations; owns (optionally) the running sums, the bicubic tap tables and a
// small pinned upload ring for the per-launch device tables.
#include "daam_types.h"
#include "../../include/daam_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

namespace daam {
hipError_t launch_tap_generic(const TapLaunch&, int, int, int, hipStream_t, int*, int*);
hipError_t launch_tap_mfma(const TapLaunch&, int acc_dtype, int max_d, int fast_exp, hipStream_t, int*, int*);
bool tap_mfma_supported(int in_dtype, int head_dim, int tokens, int hw, int64_t q_sp, int64_t k_st,
                        int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_sh);
int tap_mfma_tile_pixels();
int tap_mfma_ksteps(int head_dim);
int tap_mfma_max_steps();
hipError_t launch_tap_d64(const TapLaunch&, int in_dtype, int acc_dtype, int fast_exp, int full64, int waves8, hipStream_t, int*, int*);
int tap_d64_tile_pixels(int in_dtype, int acc_dtype, int full64);
bool tap_wide_supported(int in_dtype, int head_dim, int hw, int64_t q_sp, int64_t k_st, int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_sh,
                        const void* q, const void* k);
hipError_t launch_tap_wide(const TapLaunch&, int acc_dtype, int max_head_dim, int fast_exp, hipStream_t, int*, int*);
bool tap_d64_supported(int head_dim, int hw, int64_t q_sp, int64_t k_st, int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_sh,
                       const void* q, const void* k);
bool tap_chunk_supported(int in_dtype, int head_dim, int hw, int64_t q_sp, int64_t k_st, int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_sh,
                         int64_t q_extent, const void* q, const void* k);
hipError_t launch_tap_chunk(const TapLaunch&, int in_dtype, int acc_dtype, int fast_exp, int interleave, hipStream_t, int*, int*);
bool tap_slab_supported(int in_dtype, int batch, int heads, int head_dim, int hw, int64_t q_sp, int64_t k_st, int64_t q_sb, int64_t q_sh, int64_t k_sb,
                        int64_t k_sh, int64_t q_extent, const void* q, const void* k);
int tap_slab_heads(int head_dim);
int tap_slab_tile_pixels();
hipError_t launch_tap_slab(const TapLaunch&, int acc_dtype, int fast_exp, hipStream_t, int*, int*);

hipError_t launch_tap_probs(const ProbsLaunch&, int, int, hipStream_t, int*, int*);
hipError_t launch_finalize(const FinLaunch&, int, hipStream_t, int*, int*);
hipError_t launch_upload(void* dst, const void* src_host_mapped, size_t bytes, void* zero, size_t zero_bytes, hipStream_t);
hipError_t launch_finalize_up32_same(const FinLaunch& up, const FinLaunch& same, hipStream_t, int*);
hipError_t launch_finalize_up32_pipe(const FinPipeLaunch&, int acc_dtype, hipStream_t, int*);
int finalize_pipe_ring(int acc_dtype);
hipError_t launch_finalize_same(const FinLaunch&, int, hipStream_t, int*);
hipError_t launch_finalize_up(const FinLaunch&, int side, int, int mfma_ok, hipStream_t, int*);
bool finalize_up_supported(int side, int out_side);
bool finalize_down2_supported(int side, int out_side);
hipError_t launch_finalize_down2(const FinLaunch&, int, hipStream_t, int*);
hipError_t launch_normalize(float*, int, int, hipStream_t);
hipError_t launch_mask_overlap(const float*, int, int, const float*, int, int, int, float*, hipStream_t);
bool attend_d64_supported(int in_dtype, int head_dim, int tokens, const int64_t* strides, int n_strides, const void* const* ptrs, int n_ptrs);
hipError_t launch_attend_d64(const AttendLaunch&, int in_dtype, int acc_dtype, int fast_exp, hipStream_t, int*, int*);
hipError_t launch_clock_monitor(unsigned long long* samples, int n_samples, int period_us, hipStream_t);
hipError_t launch_start_gate(const unsigned* counter, unsigned target, int timeout_us, unsigned* timeouts, hipStream_t);
constexpr int kClockMaxSamples = 4096;
hipError_t launch_word(const float*, int, const int32_t*, int, float*, float*, int, int, int, float, float*,
                       hipStream_t);
}  // namespace daam

using namespace daam;

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) return fail((int)_e, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

namespace {

// ---- pinned upload ring -----------------------------------------------------------------
// alloc() hands out a pinned host region and its device twin; commit() copies it H2D on the
// stream with a tiny copy KERNEL that reads the (device-mapped) pinned buffer - hipMemcpyAsync put a
// ~0.2 ms cross-queue bubble between the copy and the consuming kernel - and, after the consuming
// kernel has been enqueued, release() records an event so the region is only reused once that
// kernel has run.  Wrap-around waits (hipEventSynchronize)
// only if the GPU is more than one ring behind the host.
struct Ring {
    static constexpr size_t kBytes = 8u << 20;
    char* host = nullptr;
    char* host_dev = nullptr;         // device-side address of the pinned buffer
    char* dev = nullptr;
    size_t head = 0;                  // next free byte
    struct Busy { size_t begin, end; hipEvent_t ev; };
    std::deque<Busy> busy;
    std::vector<hipEvent_t> pool;

    hipError_t init() {
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&host), kBytes, hipHostMallocMapped);
        if (e != hipSuccess) return e;
        e = hipHostGetDevicePointer(reinterpret_cast<void**>(&host_dev), host, 0);
        if (e != hipSuccess) return e;
        return hipMalloc(reinterpret_cast<void**>(&dev), kBytes);
    }
    void destroy() {
        for (auto& b : busy) { (void)hipEventSynchronize(b.ev); (void)hipEventDestroy(b.ev); }
        for (auto ev : pool) (void)hipEventDestroy(ev);
        busy.clear(); pool.clear();
        if (host) (void)hipHostFree(host);
        if (dev) (void)hipFree(dev);
        host = dev = nullptr;
    }
    bool overlaps(size_t b, size_t e) const {
        for (auto& x : busy) if (b < x.end && x.begin < e) return true;
        return false;
    }
    hipError_t alloc(size_t bytes, size_t* off) {
        bytes = (bytes + 255) & ~size_t(255);
        if (bytes > kBytes) return hipErrorOutOfMemory;
        if (head + bytes > kBytes) head = 0;
        // retire finished regions; block on the oldest ones still overlapping the request
        while (!busy.empty() && (hipEventQuery(busy.front().ev) == hipSuccess)) {
            pool.push_back(busy.front().ev);
            busy.pop_front();
        }
        while (overlaps(head, head + bytes)) {
            hipError_t e = hipEventSynchronize(busy.front().ev);
            if (e != hipSuccess) return e;
            pool.push_back(busy.front().ev);
            busy.pop_front();
        }
        *off = head;
        head += bytes;
        cur_begin = *off;
        cur_end = head;
        return hipSuccess;
    }
    size_t cur_begin = 0, cur_end = 0;
    hipError_t commit(size_t off, size_t bytes, hipStream_t s, void* zero = nullptr, size_t zero_bytes = 0) {
        return launch_upload(dev + off, host_dev + off, bytes, zero, zero_bytes, s);
    }
    hipError_t release_range(size_t begin, size_t end, hipStream_t s) {
        cur_begin = begin;
        cur_end = end;
        return release(s);
    }
    hipError_t release(hipStream_t s) {
        hipEvent_t ev;
        if (!pool.empty()) { ev = pool.back(); pool.pop_back(); }
        else {
            hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        hipError_t e = hipEventRecord(ev, s);
        if (e != hipSuccess) return e;
        busy.push_back({cur_begin, cur_end, ev});
        return hipSuccess;
    }
};

struct Layer {
    bool configured = false;
    int heads = 0, side = 0, hw = 0, factor = 0;
    void* acc = nullptr;
    bool owned = false;
    size_t bytes = 0;
    int tab = -1;
    bool dirty = false;      // tapped since the last reset (else the sums are known to be zero)
    bool zero_pending = false;  // reset() was called but the buffer has not been cleared yet: the next
                                // MFMA tap overwrites it (fresh), anything else clears it first
};

struct Pending {
    int layer;
    const void* q;
    const void* k;
    DaamQKDesc d;
};

constexpr int kMaxTabs = 16;

}  // namespace

struct DaamCtx {
    int device = 0;                    // HIP device the context was created on; every entry point runs there
    int max_layers, tokens, out_side, acc_dtype;
    std::vector<Layer> layers;
    Ring ring;
    int16_t* d_tab_idx = nullptr;
    float* d_tab_w = nullptr;
    std::vector<int> tab_sides;
    std::vector<int> tab_fp16_exact;   // every (border-merged) tap weight is an fp16 number
    void* d_up32_ops = nullptr;        // finalize_up32_mfma_kernel operands of the 32 -> 64 table (see build_up32_ops)
    void* d_up32_ops_bf16 = nullptr;   // the same for bf16 planes on the pipelined kernel: pass-1 pieces as bf16 bit patterns, W = W' + E (NULL: no such split)
    int up32_tab = -1;
    int no_mfma_finalize = 0;
    int no_fold_same = 0;             // debugging / A-B: the same-size class as its own kernel beside the pipelined one
    int no_pipe_finalize = 0;         // debugging / A-B: the round-2 x2 MFMA kernel instead of the software-pipelined one
    void* d_zero_planes = nullptr;    // [tokens][32 x 32] zeros (sized for f32 planes): padding keys of the pipelined x2 finalize
    int no_paired_finalize = 0;       // debugging / A-B: same-size and x2 class as two launches
    // finalize tables kept on the device between calls: a generation's compute_global_heat_map() selects the same keys at the same
    // addresses as the previous one, so the key / pointer tables are uploaded once and compared on the host afterwards
    static constexpr size_t kFinTabCap = 1u << 20;
    char* d_fin_tab = nullptr;
    std::vector<char> fin_tab_host;   // the bytes d_fin_tab holds (when fin_tab_valid)
    bool fin_tab_valid = false;
    hipStream_t fin_tab_stream = nullptr;   // the stream its upload and its readers were enqueued on
    int no_w8 = 0;                    // debugging / A-B: DAAM_TAP_W8=0 (head_dim-64 launches on 4-wave workgroups of 128 pixels instead of 8-wave / 256)
    int no_fin_cache = 0;             // debugging / A-B: DAAM_NO_FIN_CACHE=1 (tables through the ring + zeroing in every call)
    // daam_finalize_prepare: the output buffer the next daam_finalize accumulates into has been zeroed already (prep_*), or is to
    // be zeroed by the table-upload kernel of the next tap launch (fold_*)
    float* prep_out = nullptr;
    hipStream_t prep_stream = nullptr;
    int prep_rows = 0;                 // token rows the announced call covers (= the rows that were / will be cleared)
    float* fold_out = nullptr;
    size_t fold_bytes = 0;
    hipStream_t fold_stream = nullptr;
    std::vector<Pending> pending;
    std::vector<int> pending_count;   // per layer: recorded steps
    std::vector<int> pending_last;    // per layer: index of its newest entry in `pending`
    void drop_pending() { pending.clear(); pending_count.clear(); pending_last.clear(); }
    int last_grid[2] = {0, 0}, last_block[2] = {0, 0}, last_lds[2] = {0, 0};
    std::string last_kernels[2];       // daam_last_kernels: what the last tap launch / finalize call launched, '+'-separated
    int last_fin_side = 0;             // finalize class kernels of the last call that ran on auxiliary streams
    int last_flush_kernels = 0, last_flush_side = 0, last_flush_steps = 0;   // daam_last_flush: kernels / of them on side streams / longest step chain
    long long n_flushes = 0;           // tap launches (flushes that launched something) since the context was created
    int profile = 0;
    hipEvent_t prof_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    // daam_profile_enable(ctx, 2): every launch of a kind (0 tap, 1 finalize) gets its own event pair out of a ring, so that a caller
    // can time the launches of a whole timed region WITHOUT synchronising inside it (daam_profile_history afterwards)
    static constexpr int kProfHist = 256;
    std::vector<hipEvent_t> hist_ev[2][2];
    long long hist_count[2] = {0, 0};
    hipEvent_t prof_event(int which, int end) {
        if (profile == 2 && !hist_ev[which][end].empty()) return hist_ev[which][end][(size_t)(hist_count[which] % kProfHist)];
        return prof_ev[which][end];
    }
    // shader-clock monitor (daam_clock_monitor_*): one wave on its own stream samples the shader-cycle counter and the
    // 100 MHz reference counter into pinned memory while the kernels under test run
    unsigned long long* clk_host = nullptr;
    unsigned long long* clk_dev = nullptr;
    int clk_samples = 0;
    hipStream_t clk_stream = nullptr;
    static constexpr int kAux = 3;     // side streams of multi-kind tap flushes (see daam_tap_flush)
    hipStream_t aux_stream[kAux] = {nullptr, nullptr, nullptr};
    hipEvent_t aux_fork = nullptr, aux_join[kAux] = {nullptr, nullptr, nullptr};
    unsigned* d_started = nullptr;     // start gate of multi-kernel flushes: workgroups of side kernels started so far (wraps)
    unsigned started_target = 0;       // ... and how many the host has launched
    unsigned* gate_timeouts = nullptr; // pinned, device-mapped: gates that gave up after their 200 us (the side kernels were NOT running beside them)
    unsigned* gate_timeouts_dev = nullptr;
    int no_start_gate = 0;             // DAAM_NO_START_GATE=1 (debugging / A-B), or a failed flush left counter and target in disagreement
    unsigned gate_timeouts_seen = 0;   // value of *gate_timeouts when the gate was last armed
    unsigned long long gate_enqueued = 0;   // gated flushes enqueued since then
    long long gate_off_until = 0;      // n_flushes at which a gate that was dropped for timing out is tried again (0 = in use)
    bool gate_said = false;
    int no_side_stream = 0;

    int force_generic = 0;
    int fast_exp = 0;
    int no_d64 = 0;
    int slab_tail_pct = 25;           // DAAM_SLAB_TAIL: percent of a head_dim-40 layer's pixels the slab kernel takes in 16-pixel tiles at the end of the launch
                                      // (SD-v1.5, alternating on one box: 0 -> 2390, 25 -> 2415, 50 -> 2316, 100 -> 2204 maps/s: half-size units cost K traffic)
    int tap_slab = 1;                 // tap_slab_kernel (daam_tap_slab.hip): deferred fp16 layers of head_dim 40 / 80 / 160 in 640-byte slabs of adjacent heads
                                      // (whole 128-byte lines of Q: SD-v1.x); DAAM_TAP_SLAB=0 leaves them to the kernels below
    int tap_chunked = 2;              // tap_chunk_kernel (fp16 layers of any head_dim, one kind of workgroup): 2 = for deferred launches that
                                      // mix head dims (default), 1 = for every fp16 layer (DAAM_TAP_CHUNKED=1), 0 = never (DAAM_TAP_CHUNKED=0)
};

// Entry points may be called with another device current (a pipeline on cuda:1 while the process default is
// cuda:0): streams, events and launches must go to the context's device.  Restores the caller's device on exit.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const DaamCtx* c) {
        if (c && hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

static size_t acc_elem(int dtype) { return dtype == DAAM_F32 ? 4 : 2; }

// which running-sum dtypes a pipeline dtype may feed: its own (the reference's behaviour) or f32
static bool dtypes_compatible(int in_dtype, int acc_dtype) { return acc_dtype == DAAM_F32 || acc_dtype == in_dtype; }

// torch upsample_bicubic2d, align_corners=False, antialias=False:
// scale = in / out in f32; src = scale * (dst + 0.5) - 0.5 (NOT clamped for cubic);
// taps floor(src)-1 .. +2 clamped to the border; A = -0.75.
static void bicubic_table(int in_size, int out_size, int16_t* idx, float* w)
{
#pragma clang fp contract(off)
    const float A = -0.75f;
    const float scale = (float)in_size / (float)out_size;
    for (int j = 0; j < out_size; ++j) {
        const float src = scale * ((float)j + 0.5f) - 0.5f;
        const float f = std::floor(src);
        const float t = src - f;
        const float x0 = t + 1.0f, u = 1.0f - t, x3 = u + 1.0f;
        w[j * 4 + 0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
        w[j * 4 + 1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
        w[j * 4 + 2] = ((A + 2.0f) * u - (A + 3.0f)) * u * u + 1.0f;
        w[j * 4 + 3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
        for (int a = 0; a < 4; ++a) {
            int v = (int)f - 1 + a;
            v = v < 0 ? 0 : (v > in_size - 1 ? in_size - 1 : v);
            idx[j * 4 + a] = (int16_t)v;
        }
    }
}

static int ensure_zeroed(Layer& l, hipStream_t s)
{
    if (l.zero_pending) {
        HIP_TRY(hipMemsetAsync(l.acc, 0, l.bytes, s));
        l.zero_pending = false;
    }
    return 0;
}

// MFMA operand pieces of finalize_up32_mfma_kernel (32 -> 64, fp16-exact banded tap matrix W[o][src]),
// one 16-byte piece per (nt, lane, k): lane = (n = lane & 31, g = lane >> 5)
//   k = 0, 1      : Wx, B of pass 1:  W[32nt + n][16ks + 8g + e],                      ks = k
//   k = 2 + 2t+ks : Wy, A of pass 2:  W[32t + n][16ks + 8(i >> 2) + 4g + (i & 3)]     (contraction index
//                   permuted to the C/D register order of pass 1, see the kernel)
// wx_bf16 (bf16 planes on the pipelined kernel, pass 1 on the bf16 MFMA): 7 pieces per (nt, lane) -- k = 0, 1 hold W' as bf16 bit patterns and
//   k = 6         : E, B of the third pass-1 MFMA:  E[32nt + n][g == 0 ? e : 24 + e]    with W = W' + E, both bf16 numbers
// (W' = W truncated to eight significant bits; the 32 -> 64 table has one entry per border that needs nine: 283/256 = 282/256 + 1/256).
// *ok = 0 when E is not a bf16 number somewhere or has an entry outside source columns 0..7 / 24..31.
static std::vector<_Float16> build_up32_ops(const int16_t* idx, const float* w, bool wx_bf16 = false, int* ok = nullptr)
{
    auto W = [&](int o, int src) {
        float v = 0.f;
        for (int a = 0; a < 4; ++a)
            if (idx[o * 4 + a] == src) v += w[o * 4 + a];
        return (_Float16)v;
    };
    auto trunc_bf16 = [](float v) {
        uint32_t u;
        memcpy(&u, &v, 4);
        u &= 0xffff0000u;
        memcpy(&v, &u, 4);
        return v;
    };
    auto put_bf16 = [](_Float16* dst, float v) {
        const uint16_t bits = f32_to_bf16(v).bits;
        memcpy(dst, &bits, 2);
    };
    const int pieces = wx_bf16 ? 7 : 6;
    if (ok) *ok = 1;
    std::vector<_Float16> ops((size_t)2 * 64 * pieces * 8);
    for (int nt = 0; nt < 2; ++nt)
        for (int lane = 0; lane < 64; ++lane) {
            const int n = lane & 31, g = lane >> 5;
            _Float16* dst = ops.data() + ((size_t)(nt * 64 + lane) * pieces) * 8;
            for (int ks = 0; ks < 2; ++ks)
                for (int e = 0; e < 8; ++e) {
                    const int src = 16 * ks + 8 * g + e;
                    const float v = (float)W(32 * nt + n, src);
                    if (!wx_bf16) { dst[ks * 8 + e] = (_Float16)v; continue; }
                    const float hi = trunc_bf16(v), lo = v - hi;
                    put_bf16(&dst[ks * 8 + e], hi);
                    if (lo != 0.f && ok && (bf16_to_f32(f32_to_bf16(lo)) != lo || (src >= 8 && src < 24))) *ok = 0;
                }
            for (int t = 0; t < 2; ++t)
                for (int ks = 0; ks < 2; ++ks)
                    for (int i = 0; i < 8; ++i)
                        dst[(2 + 2 * t + ks) * 8 + i] = W(32 * t + n, 16 * ks + 8 * (i >> 2) + 4 * g + (i & 3));
            if (wx_bf16)
                for (int e = 0; e < 8; ++e) {
                    const float v = (float)W(32 * nt + n, g == 0 ? e : 24 + e);
                    put_bf16(&dst[6 * 8 + e], v - trunc_bf16(v));
                }
        }
    return ops;
}

// Key ranges of the chunks of the x2 MFMA finalize: equal shares (even boundaries; the two key lanes of a workgroup take the
// keys of its range alternately).  Shares shrinking with the dispatch round of a chunk's workgroups (the SIMD arbitrates by
// age: the first 256 workgroups finish their loop in 23 us, the last 256 in 40 us) were tried and changed nothing -- the kernel
// is throughput-bound from its first to its last microsecond, the age order only decides who waits.
static void finalize_chunk_ranges(int n_keys, int n_chunks, FinLaunch* L)
{
    const int pairs = (n_keys + 1) / 2;
    for (int c = 0; c <= n_chunks; ++c)
        L->chunk_begin[c] = (int16_t)std::min(n_keys, 2 * (int)(((int64_t)pairs * c + n_chunks - 1) / n_chunks));
    L->chunk_begin[n_chunks] = (int16_t)n_keys;
}

// auxiliary non-blocking streams + fork / join events of a context (multi-kernel tap flushes, multi-class finalize)
static hipError_t ensure_aux(DaamCtx* c)
{
    if (c->aux_fork) return hipSuccess;
    hipError_t ae = hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming);
    for (int i = 0; i < DaamCtx::kAux && ae == hipSuccess; ++i) {
        ae = hipStreamCreateWithFlags(&c->aux_stream[i], hipStreamNonBlocking);
        if (ae == hipSuccess) ae = hipEventCreateWithFlags(&c->aux_join[i], hipEventDisableTiming);
    }
    if (ae == hipSuccess && !c->d_started) {
        ae = hipMalloc(reinterpret_cast<void**>(&c->d_started), sizeof(unsigned));
        if (ae == hipSuccess) ae = hipMemset(c->d_started, 0, sizeof(unsigned));
        c->started_target = 0;
    }
    if (ae == hipSuccess && !c->gate_timeouts) {
        // best effort: without the word the gate still works, its timeouts just go unnoticed
        if (hipHostMalloc(reinterpret_cast<void**>(&c->gate_timeouts), sizeof(unsigned), hipHostMallocMapped) == hipSuccess) {
            *c->gate_timeouts = 0;
            if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->gate_timeouts_dev), c->gate_timeouts, 0) != hipSuccess)
                c->gate_timeouts_dev = nullptr;
        } else {
            c->gate_timeouts = nullptr;
        }
    }
    return ae;
}

static const char* tap_kernel_name(int kd)
{
    return (kd == 65 || kd == 66) ? "tap_d64_kernel" : (kd == 67 || kd == 69) ? "tap_wide_kernel" : kd == 70 ? "tap_chunk_kernel"
           : kd == 71 ? "tap_slab_kernel" : kd ? "tap_mfma_kernel" : "tap_generic_kernel";
}
static const char* dtype_name(int dt) { return dt == DAAM_F32 ? "f32" : dt == DAAM_BF16 ? "bf16" : "f16"; }

extern "C" {

int daam_abi_version(void) { return DAAM_ABI_VERSION; }
const char* daam_last_error(void) { return g_err.c_str(); }

int daam_ctx_create(int max_layers, int tokens, int out_side, int acc_dtype, DaamCtx** out)
{
    if (!out) return fail(DAAM_E_INVALID, "out is NULL");
    *out = nullptr;
    if (max_layers <= 0 || max_layers > 4096) return fail(DAAM_E_INVALID, "max_layers %d out of range", max_layers);
    if (tokens <= 0 || tokens > kMaxTokens) return fail(DAAM_E_INVALID, "tokens %d not in 1..%d", tokens, kMaxTokens);
    if (out_side <= 0 || out_side > 128) return fail(DAAM_E_INVALID, "out_side %d not in 1..128", out_side);
    if (acc_dtype != DAAM_F16 && acc_dtype != DAAM_F32 && acc_dtype != DAAM_BF16) return fail(DAAM_E_INVALID, "acc_dtype %d", acc_dtype);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail((int)hipErrorNoDevice, "no HIP device");
    DaamCtx* c = new DaamCtx();
    HIP_TRY(hipGetDevice(&c->device));
    c->max_layers = max_layers;
    c->tokens = tokens;
    c->out_side = out_side;
    c->acc_dtype = acc_dtype;
    c->layers.resize(max_layers);
    hipError_t e = c->ring.init();
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_tab_idx), sizeof(int16_t) * kMaxTabs * out_side * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_tab_w), sizeof(float) * kMaxTabs * out_side * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_fin_tab), DaamCtx::kFinTabCap);
    if (e != hipSuccess) {
        daam_ctx_destroy(c);
        return fail((int)e, "context allocation: %s", hipGetErrorString(e));
    }
    const char* fg = getenv("DAAM_FORCE_GENERIC");
    c->force_generic = fg && fg[0] == '1';
    const char* nm = getenv("DAAM_NO_MFMA_FINALIZE");
    c->no_mfma_finalize = nm && nm[0] == '1';
    const char* nfs = getenv("DAAM_NO_FOLD_SAME");
    c->no_fold_same = nfs && nfs[0] == '1';
    const char* nsg = getenv("DAAM_NO_START_GATE");
    c->no_start_gate = nsg && nsg[0] == '1';
    const char* npp = getenv("DAAM_NO_PIPE_FINALIZE");
    c->no_pipe_finalize = npp && npp[0] == '1';
    const char* npf = getenv("DAAM_NO_PAIRED_FINALIZE");
    c->no_paired_finalize = npf && npf[0] == '1';
    const char* w8 = getenv("DAAM_TAP_W8");
    c->no_w8 = w8 && w8[0] == '0';
    const char* nfc = getenv("DAAM_NO_FIN_CACHE");
    c->no_fin_cache = nfc && nfc[0] == '1';
    const char* n16 = getenv("DAAM_NO_D64");            // debugging: 32x32-tile kernel also for head_dim 64
    c->no_d64 = n16 && n16[0] == '1';
    const char* nss = getenv("DAAM_NO_SIDE_STREAM");        // debugging / A-B: every tap kernel of a flush on the caller's stream
    c->no_side_stream = nss && nss[0] == '1';
    const char* stl = getenv("DAAM_SLAB_TAIL");
    if (stl && stl[0]) c->slab_tail_pct = std::max(0, std::min(100, atoi(stl)));
    const char* tsl = getenv("DAAM_TAP_SLAB");
    c->tap_slab = !(tsl && tsl[0] == '0');
    const char* tck = getenv("DAAM_TAP_CHUNKED");           // daam_tap_chunk.hip: unset = launches that mix head dims, 1 = always, 0 = never
    c->tap_chunked = !tck || !tck[0] ? 2 : tck[0] == '1' ? 1 : tck[0] == '0' ? 0 : 2;

    // softmax flavour of the MFMA tap: fast (default; exponent by one mixed-precision FMA, ~1e-6 relative,
    // same deviation class as the f32 summation order of q.k -- DESIGN.md section 3.1) or compensated
    // (DAAM_STRICT_EXP=1: ~1 ulp f32 like the reference's expf)
    const char* se = getenv("DAAM_STRICT_EXP");
    c->fast_exp = !(se && se[0] == '1');
    *out = c;
    return 0;
}

int daam_ctx_destroy(DaamCtx* c)
{
    if (!c) return 0;
    DeviceGuard on_device(c);
    (void)hipDeviceSynchronize();
    c->ring.destroy();
    for (auto& l : c->layers)
        if (l.owned && l.acc) (void)hipFree(l.acc);
    for (auto& pair : c->prof_ev)
        for (auto& ev : pair)
            if (ev) (void)hipEventDestroy(ev);
    for (auto& pair : c->hist_ev)
        for (auto& ring : pair)
            for (auto ev : ring)
                if (ev) (void)hipEventDestroy(ev);
    if (c->aux_fork) (void)hipEventDestroy(c->aux_fork);
    if (c->d_started) (void)hipFree(c->d_started);
    if (c->gate_timeouts) (void)hipHostFree(c->gate_timeouts);
    for (auto& ev : c->aux_join)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& st : c->aux_stream)
        if (st) (void)hipStreamDestroy(st);
    if (c->clk_stream) (void)hipStreamDestroy(c->clk_stream);
    if (c->clk_host) (void)hipHostFree(c->clk_host);
    if (c->d_up32_ops) (void)hipFree(c->d_up32_ops);
    if (c->d_up32_ops_bf16) (void)hipFree(c->d_up32_ops_bf16);
    if (c->d_zero_planes) (void)hipFree(c->d_zero_planes);
    if (c->d_fin_tab) (void)hipFree(c->d_fin_tab);
    if (c->d_tab_idx) (void)hipFree(c->d_tab_idx);
    if (c->d_tab_w) (void)hipFree(c->d_tab_w);
    delete c;
    return 0;
}

int daam_layer_configure(DaamCtx* c, int layer, int heads, int side, int factor, void* acc)
{
    if (!c) return fail(DAAM_E_INVALID, "ctx is NULL");
    if (layer < 0 || layer >= c->max_layers) return fail(DAAM_E_INVALID, "layer %d out of range", layer);
    if (heads <= 0 || side <= 0 || side > 1024) return fail(DAAM_E_INVALID, "heads %d / side %d", heads, side);
    DeviceGuard on_device(c);
    for (auto& p : c->pending)
        if (p.layer == layer) return fail(DAAM_E_STATE, "layer %d re-configured with un-flushed taps pending", layer);
    Layer& l = c->layers[layer];
    if (l.owned && l.acc) { HIP_TRY(hipFree(l.acc)); }
    l = Layer();
    l.heads = heads;
    l.side = side;
    l.hw = side * side;
    l.factor = factor;
    l.bytes = (size_t)heads * c->tokens * l.hw * acc_elem(c->acc_dtype);
    if (acc) {
        l.acc = acc;
    } else {
        HIP_TRY(hipMalloc(&l.acc, l.bytes));
        HIP_TRY(hipMemset(l.acc, 0, l.bytes));
        l.owned = true;
    }
    if (side != c->out_side) {
        int tab = -1;
        for (size_t i = 0; i < c->tab_sides.size(); ++i)
            if (c->tab_sides[i] == side) tab = (int)i;
        if (tab < 0) {
            if ((int)c->tab_sides.size() >= kMaxTabs) return fail(DAAM_E_UNSUPPORTED, "more than %d distinct map sizes", kMaxTabs);
            tab = (int)c->tab_sides.size();
            std::vector<int16_t> idx(c->out_side * 4);
            std::vector<float> w(c->out_side * 4);
            bicubic_table(side, c->out_side, idx.data(), w.data());
            HIP_TRY(hipMemcpy(c->d_tab_idx + (size_t)tab * c->out_side * 4, idx.data(), idx.size() * sizeof(int16_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->d_tab_w + (size_t)tab * c->out_side * 4, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
            c->tab_sides.push_back(side);
            // can the banded tap matrix be an fp16 MFMA operand without rounding?
            int exact = 1;
            for (int j = 0; j < c->out_side && exact; ++j)
                for (int a = 0; a < 4 && exact; ++a) {
                    float merged = 0.f;                    // taps clamped onto the same border column add up
                    for (int b2 = 0; b2 < 4; ++b2)
                        if (idx[j * 4 + b2] == idx[j * 4 + a]) merged += w[j * 4 + b2];
                    exact = ((float)(_Float16)merged == merged) && ((float)(_Float16)w[j * 4 + a] == w[j * 4 + a]);
                }
            c->tab_fp16_exact.push_back(exact);
            if (exact && side == 32 && c->out_side == 64 && !c->d_up32_ops) {
                std::vector<_Float16> ops = build_up32_ops(idx.data(), w.data());
                HIP_TRY(hipMalloc(&c->d_up32_ops, ops.size() * sizeof(_Float16)));
                HIP_TRY(hipMemcpy(c->d_up32_ops, ops.data(), ops.size() * sizeof(_Float16), hipMemcpyHostToDevice));
                // bf16 planes: W = W' + E as two bf16 operands (build_up32_ops checks that the table splits that way)
                int bf_ok = 0;
                ops = build_up32_ops(idx.data(), w.data(), true, &bf_ok);
                if (bf_ok) {
                    HIP_TRY(hipMalloc(&c->d_up32_ops_bf16, ops.size() * sizeof(_Float16)));
                    HIP_TRY(hipMemcpy(c->d_up32_ops_bf16, ops.data(), ops.size() * sizeof(_Float16), hipMemcpyHostToDevice));
                }
                c->up32_tab = tab;
                const size_t zb = (size_t)c->tokens * 32 * 32 * sizeof(float);
                HIP_TRY(hipMalloc(&c->d_zero_planes, zb));
                HIP_TRY(hipMemset(c->d_zero_planes, 0, zb));
            }
        }
        l.tab = tab;
    }
    l.configured = true;
    return 0;
}

int daam_layer_acc(DaamCtx* c, int layer, void** acc, size_t* bytes)
{
    if (!c || layer < 0 || layer >= c->max_layers || !c->layers[layer].configured)
        return fail(DAAM_E_STATE, "layer %d not configured", layer);
    if (acc) *acc = c->layers[layer].acc;
    if (bytes) *bytes = c->layers[layer].bytes;
    return 0;
}

int daam_layer_touch(DaamCtx* c, int layer, void* stream)
{
    if (!c || layer < 0 || layer >= c->max_layers || !c->layers[layer].configured)
        return fail(DAAM_E_STATE, "layer %d not configured", layer);
    if (!c->pending.empty()) return fail(DAAM_E_STATE, "layer touched with deferred taps pending: flush first");
    DeviceGuard on_device(c);
    Layer& l = c->layers[layer];
    int zrc = ensure_zeroed(l, (hipStream_t)stream);       // a reset still owed to the buffer happens first, on this stream
    if (zrc) return zrc;
    l.dirty = true;                                        // later taps add to the sums, the next reset clears them
    return 0;
}

int daam_layer_release(DaamCtx* c, int layer)
{
    if (!c || layer < 0 || layer >= c->max_layers) return fail(DAAM_E_INVALID, "layer %d out of range", layer);
    for (auto& p : c->pending)
        if (p.layer == layer) return fail(DAAM_E_STATE, "layer %d released with un-flushed taps pending", layer);
    Layer& l = c->layers[layer];
    if (l.owned && l.acc) {
        DeviceGuard on_device(c);
        HIP_TRY(hipFree(l.acc));
    }
    l = Layer();                                               // unconfigured: no later call touches the old buffer
    return 0;
}

int daam_reset(DaamCtx* c, void* stream)
{
    if (!c) return fail(DAAM_E_INVALID, "ctx is NULL");
    c->drop_pending();
    c->prep_out = c->fold_out = nullptr;
    (void)stream;
    for (auto& l : c->layers)
        if (l.configured) {
            // lazy: a layer that is tapped again is overwritten by its first launch (TapLayer.fresh),
            // so the 221 MB of memsets per generation are only paid for paths that read the sums first
            l.zero_pending = l.zero_pending || l.dirty;
            l.dirty = false;
        }
    return 0;
}

static int check_qk(DaamCtx* c, int layer, const void* q, const void* k, const DaamQKDesc* d)
{
    if (!c || !d || !q || !k) return fail(DAAM_E_INVALID, "NULL argument");
    if (layer < 0 || layer >= c->max_layers || !c->layers[layer].configured)
        return fail(DAAM_E_STATE, "layer %d not configured", layer);
    const Layer& l = c->layers[layer];
    if (d->in_dtype != DAAM_F16 && d->in_dtype != DAAM_F32 && d->in_dtype != DAAM_BF16) return fail(DAAM_E_INVALID, "in_dtype %d", d->in_dtype);
    if (!dtypes_compatible(d->in_dtype, c->acc_dtype))
        return fail(DAAM_E_INVALID, "activations of dtype %d cannot feed running sums of dtype %d (own dtype or f32)", d->in_dtype, c->acc_dtype);
    if (d->tokens != c->tokens) return fail(DAAM_E_INVALID, "tokens %d != context size %d (reference gate, trace.py:289)", d->tokens, c->tokens);
    if (d->batch <= 0 || d->heads <= 0 || d->head_dim <= 0 || d->head_dim > 1024)
        return fail(DAAM_E_INVALID, "batch %d heads %d head_dim %d", d->batch, d->heads, d->head_dim);
    const int bh = d->batch * d->heads;
    if (bh - bh / 2 != l.heads) return fail(DAAM_E_INVALID, "layer %d holds %d heads, call keeps %d", layer, l.heads, bh - bh / 2);
    if (d->hw != l.hw) return fail(DAAM_E_INVALID, "layer %d holds %d positions, call has %d", layer, l.hw, d->hw);
    return 0;
}

static void fill_layer(const DaamCtx* c, const Layer& l, const DaamQKDesc& d, int tile_pixels, TapLayer* t)
{
    const int bh = d.batch * d.heads;
    t->acc = l.acc;
    t->heads_kept = l.heads;
    t->bh_first = bh / 2;
    t->heads = d.heads;
    t->hw = d.hw;
    t->head_dim = d.head_dim;
    t->tiles_per_head = (d.hw + tile_pixels - 1) / tile_pixels;
    t->wg_begin = 0;
    t->n_steps = 1;
    t->ptr_begin = 0;
    t->round_logits = d.round_logits;
    t->scale = d.scale;
    t->fresh = l.dirty ? 0 : 1;
    t->px_begin = 0;
    t->px_end = d.hw;
    t->tile_px = tile_pixels;
    t->q_sb = d.q_stride_b; t->q_sh = d.q_stride_h; t->q_sp = d.q_stride_p;
    t->k_sb = d.k_stride_b; t->k_sh = d.k_stride_h; t->k_st = d.k_stride_t;
    (void)c;
}

static bool use_d64_bf16(const DaamCtx* c, const DaamQKDesc& d, const void* q, const void* k);
static bool use_chunk(const DaamCtx* c, const DaamQKDesc& d, const void* q, const void* k);

static bool use_mfma(const DaamCtx* c, const DaamQKDesc& d, const void* q, const void* k)
{
    if (c->force_generic) return false;
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15) return false;
    if (d.in_dtype == DAAM_BF16) return use_d64_bf16(c, d, q, k) || use_chunk(c, d, q, k);
    return tap_mfma_supported(d.in_dtype, d.head_dim, d.tokens, d.hw, d.q_stride_p, d.k_stride_t, d.q_stride_b,
                              d.q_stride_h, d.k_stride_b, d.k_stride_h);
}

// kernel choice for an MFMA-capable call: 65 = 16x16-tile head_dim-64 kernel (default for d = 64),
// else the k-step count of the generic 32x32-tile MFMA kernel
static int mfma_kind(const DaamCtx* c, const DaamQKDesc& d, const void* q, const void* k);

// The specialised tap kernels address Q and K with 32-bit BYTE offsets from the tensor pointers: every element a call can
// address -- (batch - 1) * stride_b + (heads - 1) * stride_h + (rows - 1) * row stride + head_dim -- must stay below 2^30 elements
// (views into a large fused buffer with unusual strides fall back to the any-shape kernel instead of wrapping).
static bool offsets_fit_32(const DaamQKDesc& d)
{
    const int64_t lim = (int64_t)1 << 30;
    const int64_t s[] = {d.q_stride_b, d.q_stride_h, d.q_stride_p, d.k_stride_b, d.k_stride_h, d.k_stride_t};
    for (int64_t v : s)
        if (v < 0 || v >= lim) return false;
    const int64_t q_max = (int64_t)(d.batch - 1) * d.q_stride_b + (int64_t)(d.heads - 1) * d.q_stride_h + (int64_t)(d.hw - 1) * d.q_stride_p + d.head_dim;
    const int64_t k_max = (int64_t)(d.batch - 1) * d.k_stride_b + (int64_t)(d.heads - 1) * d.k_stride_h + (int64_t)(d.tokens - 1) * d.k_stride_t + d.head_dim;
    return q_max < lim && k_max < lim;
}

static bool use_d64(const DaamCtx* c, const DaamQKDesc& d, const void* q, const void* k)
{
    return !c->no_d64 && offsets_fit_32(d) && tap_d64_supported(d.head_dim, d.hw, d.q_stride_p, d.k_stride_t, d.q_stride_b, d.q_stride_h,
                                           d.k_stride_b, d.k_stride_h, q, k);
}

// bf16 pipelines: only the 16x16-tile kernel has a bf16 variant (head_dim <= 64, 77 tokens, bf16-rounded logits);
// everything else of a bf16 pipeline runs on the any-shape kernel
static bool use_d64_bf16(const DaamCtx* c, const DaamQKDesc& d, const void* q, const void* k)
{
    return d.in_dtype == DAAM_BF16 && !c->no_d64 && c->fast_exp && d.round_logits && d.tokens == 77 && d.hw % 8 == 0 && offsets_fit_32(d) &&
           tap_d64_supported(d.head_dim, d.hw, d.q_stride_p, d.k_stride_t, d.q_stride_b, d.q_stride_h, d.k_stride_b,
                             d.k_stride_h, q, k);
}

// 64 < head_dim <= 160 on fp16 pipelines (SD-v1.5's 80 / 160): the 16x16-tile kernel with 3 or 5 k-steps (daam_tap_wide.hip)
static bool use_wide(const DaamCtx* c, const DaamQKDesc& d, const void* q, const void* k)
{
    return !c->no_d64 && (c->acc_dtype == DAAM_F16 || c->acc_dtype == DAAM_F32) && offsets_fit_32(d) &&
           (int64_t)d.batch * d.q_stride_b < ((int64_t)1 << 30) &&
           tap_wide_supported(d.in_dtype, d.head_dim, d.hw, d.q_stride_p, d.k_stride_t, d.q_stride_b, d.q_stride_h, d.k_stride_b,
                              d.k_stride_h, q, k);
}

// the chunked kernel (daam_tap_chunk.hip) can take this call: fp16 layers of any head_dim (multiple of 8, <= 256), and bf16 layers (bf16
// or f32 sums, bf16-rounded logits, the fast softmax -- what the bf16 head_dim-64 kernel asks for; validated on the chip in round 4)
static bool chunk_ok(const DaamCtx* c, const DaamQKDesc& d, const void* q, const void* k)
{
    if (!c->tap_chunked || c->no_d64 || c->force_generic || d.tokens != 77 || !offsets_fit_32(d)) return false;
    if (d.in_dtype == DAAM_BF16) {
        if (!c->fast_exp || !d.round_logits || !(c->acc_dtype == DAAM_BF16 || c->acc_dtype == DAAM_F32)) return false;
    } else if (!(c->acc_dtype == DAAM_F16 || c->acc_dtype == DAAM_F32)) {
        return false;
    }
    return tap_chunk_supported(d.in_dtype, d.head_dim, d.hw, d.q_stride_p, d.k_stride_t, d.q_stride_b, d.q_stride_h, d.k_stride_b,
                               d.k_stride_h, (int64_t)d.batch * d.q_stride_b, q, k);
}

// the slab kernel (daam_tap_slab.hip) can take this deferred call: fp16 Q / K, fp16 or f32 sums, head_dim 40 / 80 / 160 with the heads
// adjacent in the rows
static bool slab_ok(const DaamCtx* c, const DaamQKDesc& d, const void* q, const void* k)
{
    if (!c->tap_slab || c->no_d64 || c->force_generic || d.tokens != 77 || d.in_dtype != DAAM_F16 || !offsets_fit_32(d)) return false;
    if (!(c->acc_dtype == DAAM_F16 || c->acc_dtype == DAAM_F32)) return false;
    return tap_slab_supported(d.in_dtype, d.batch, d.heads, d.head_dim, d.hw, d.q_stride_p, d.k_stride_t, d.q_stride_b, d.q_stride_h, d.k_stride_b,
                              d.k_stride_h, (int64_t)d.batch * d.q_stride_b, q, k);
}

// DAAM_TAP_CHUNKED=1: every such call.  Default (2): the deferred launches that mix head dims (daam_tap_flush), and bf16 layers with
// head_dim > 64 -- no other MFMA kernel has a bf16 form for them (the any-shape kernel is ~45x slower per step).
static bool use_chunk(const DaamCtx* c, const DaamQKDesc& d, const void* q, const void* k)
{
    if (c->tap_chunked == 1) return chunk_ok(c, d, q, k);
    return c->tap_chunked == 2 && d.in_dtype == DAAM_BF16 && d.head_dim > 64 && chunk_ok(c, d, q, k);
}

static int mfma_kind(const DaamCtx* c, const DaamQKDesc& d, const void* q, const void* k)
{
    if (use_chunk(c, d, q, k)) return 70;
    if (d.in_dtype == DAAM_BF16) return 66;                   // only reached when use_d64_bf16() holds
    if (use_d64(c, d, q, k)) return 65;
    if (use_wide(c, d, q, k)) return d.head_dim <= 96 ? 67 : 69;
    return tap_mfma_ksteps(d.head_dim);
}

int daam_tap_qk(DaamCtx* c, int layer, const void* q, const void* k, const DaamQKDesc* d, void* stream)
{
    int rc = check_qk(c, layer, q, k, d);
    if (rc) return rc;
    if (!c->pending.empty()) return fail(DAAM_E_STATE, "immediate tap with deferred taps pending: flush first");
    DeviceGuard on_device(c);
    const bool mfma = use_mfma(c, *d, q, k);
    if (!mfma && (rc = ensure_zeroed(c->layers[layer], (hipStream_t)stream))) return rc;
    TapLaunch L;
    memset(&L, 0, sizeof L);
    fill_layer(c, c->layers[layer], *d, mfma ? tap_mfma_tile_pixels() : kTapPixels, &L.one);
    L.one_ptr.q = q;
    L.one_ptr.k = k;
    L.n_layers = 1;
    L.tokens = c->tokens;
    L.total_wgs = L.one.heads_kept * L.one.tiles_per_head;
    L.wgs_per_xcd = (L.total_wgs + 7) / 8;
    c->last_block[0] = 256;
    const int kd1 = mfma ? mfma_kind(c, *d, q, k) : 0;
    hipError_t e = (kd1 == 65 || kd1 == 66) ? launch_tap_d64(L, d->in_dtype, c->acc_dtype, c->fast_exp && d->round_logits, d->head_dim == 64, 0, (hipStream_t)stream, &c->last_grid[0], &c->last_lds[0])
                   : (kd1 == 67 || kd1 == 69) ? launch_tap_wide(L, c->acc_dtype, d->head_dim, c->fast_exp && d->round_logits, (hipStream_t)stream, &c->last_grid[0], &c->last_lds[0])
                   : kd1 == 70 ? launch_tap_chunk(L, d->in_dtype, c->acc_dtype, c->fast_exp && d->round_logits, 0, (hipStream_t)stream, &c->last_grid[0], &c->last_lds[0])
                   : mfma ? launch_tap_mfma(L, c->acc_dtype, d->head_dim, c->fast_exp && d->round_logits, (hipStream_t)stream, &c->last_grid[0], &c->last_lds[0])
                        : launch_tap_generic(L, d->in_dtype, c->acc_dtype, d->head_dim, (hipStream_t)stream,
                                             &c->last_grid[0], &c->last_lds[0]);
    if (e != hipSuccess) return fail((int)e, "tap launch: %s", hipGetErrorString(e));
    c->last_kernels[0] = tap_kernel_name(kd1);
    c->layers[layer].dirty = true;
    c->layers[layer].zero_pending = false;
    return 0;
}

int daam_attend_supported(const DaamAttendDesc* d, const void* q, const void* k, const void* v, const void* out)
{
    if (!d || !q || !k || !v || !out) return 0;
    const int64_t strides[] = {d->qk.q_stride_b, d->qk.q_stride_h, d->qk.q_stride_p, d->qk.k_stride_b, d->qk.k_stride_h,
                               d->qk.k_stride_t, d->v_stride_b, d->v_stride_h, d->v_stride_t, d->o_stride_b, d->o_stride_h,
                               d->o_stride_p};
    const void* ptrs[] = {q, k, v, out};
    // hw % 8: the fused tap updates the running sums in 16-byte row pieces (8 fp16 pixels), rows must start aligned
    if (d->qk.batch <= 0 || d->qk.heads <= 0 || d->qk.hw <= 0 || d->qk.hw % 8 != 0) return 0;
    return attend_d64_supported(d->qk.in_dtype, d->qk.head_dim, d->qk.tokens, strides, 12, ptrs, 4) ? 1 : 0;
}

int daam_attend(DaamCtx* c, int layer, const void* q, const void* k, const void* v, void* out, const DaamAttendDesc* d,
                int tap, void* stream)
{
    if (!c || !d || !q || !k || !v || !out) return fail(DAAM_E_INVALID, "NULL argument");
    if (!daam_attend_supported(d, q, k, v, out))
        return fail(DAAM_E_UNSUPPORTED, "daam_attend: fp16 / bf16, head_dim %% 8 == 0 up to 160, 77 tokens, hw %% 8 == 0, strides %% 8 == 0, 16-byte aligned pointers only");
    if (d->qk.in_dtype == DAAM_BF16 && !d->qk.round_logits)
        return fail(DAAM_E_UNSUPPORTED, "daam_attend: bf16 pipelines with upcast_attention (f32 logits) take the framework's attention");
    if (!dtypes_compatible(d->qk.in_dtype, c->acc_dtype))
        return fail(DAAM_E_UNSUPPORTED, "daam_attend: activations of dtype %d on a context whose sums are dtype %d", d->qk.in_dtype, c->acc_dtype);
    if (d->qk.tokens != c->tokens) return fail(DAAM_E_INVALID, "tokens %d != context size %d", d->qk.tokens, c->tokens);
    if (tap) {
        int rc = check_qk(c, layer, q, k, &d->qk);
        if (rc) return rc;
        if (!c->pending.empty()) return fail(DAAM_E_STATE, "fused tap with deferred taps pending: flush first");
    }
    DeviceGuard on_device(c);
    AttendLaunch L;
    memset(&L, 0, sizeof L);
    L.q = q; L.k = k; L.v = v; L.out = out;
    L.batch = d->qk.batch; L.heads = d->qk.heads; L.hw = d->qk.hw; L.head_dim = d->qk.head_dim;
    L.tiles_per_head = (d->qk.hw + tap_mfma_tile_pixels() - 1) / tap_mfma_tile_pixels();
    L.total_wgs = L.batch * L.heads * L.tiles_per_head;
    L.wgs_per_xcd = (L.total_wgs + 7) / 8;
    L.bh_first = (d->qk.batch * d->qk.heads) / 2;
    L.round_logits = d->qk.round_logits;
    L.scale = d->qk.scale;
    L.q_sb = d->qk.q_stride_b; L.q_sh = d->qk.q_stride_h; L.q_sp = d->qk.q_stride_p;
    L.k_sb = d->qk.k_stride_b; L.k_sh = d->qk.k_stride_h; L.k_st = d->qk.k_stride_t;
    L.v_sb = d->v_stride_b; L.v_sh = d->v_stride_h; L.v_st = d->v_stride_t;
    L.o_sb = d->o_stride_b; L.o_sh = d->o_stride_h; L.o_sp = d->o_stride_p;
    if (tap) {
        L.acc = c->layers[layer].acc;
        L.fresh = c->layers[layer].dirty ? 0 : 1;
    }
    int grid = 0, lds = 0;
    hipError_t e = launch_attend_d64(L, d->qk.in_dtype, c->acc_dtype, c->fast_exp && d->qk.round_logits, (hipStream_t)stream, &grid, &lds);
    if (e != hipSuccess) return fail((int)e, "attend launch: %s", hipGetErrorString(e));
    if (tap) {
        c->last_grid[0] = grid;
        c->last_block[0] = 256;
        c->last_lds[0] = lds;
        c->layers[layer].dirty = true;
        c->layers[layer].zero_pending = false;
    }
    return 0;
}

int daam_tap_qk_enqueue(DaamCtx* c, int layer, const void* q, const void* k, const DaamQKDesc* d)
{
    int rc = check_qk(c, layer, q, k, d);
    if (rc) return rc;
    if (!c->pending.empty() && c->pending.front().d.in_dtype != d->in_dtype)
        return fail(DAAM_E_STATE, "mixed activation dtypes in one deferred batch: flush first");
    if (c->pending_count.size() != (size_t)c->max_layers) {
        c->pending_count.assign(c->max_layers, 0);
        c->pending_last.assign(c->max_layers, -1);
    }
    if (c->pending_count[layer] > 0) {
        // every recorded step of a layer must share shape and strides
        const DaamQKDesc& o = c->pending[c->pending_last[layer]].d;
        if (o.batch != d->batch || o.heads != d->heads || o.head_dim != d->head_dim ||
            o.round_logits != d->round_logits || o.scale != d->scale || o.q_stride_b != d->q_stride_b ||
            o.q_stride_h != d->q_stride_h || o.q_stride_p != d->q_stride_p || o.k_stride_b != d->k_stride_b ||
            o.k_stride_h != d->k_stride_h || o.k_stride_t != d->k_stride_t)
            return fail(DAAM_E_STATE, "layer %d changed shape inside a deferred batch: flush first", layer);
        if (c->pending_count[layer] >= tap_mfma_max_steps())
            return fail(DAAM_E_STATE, "layer %d already has %d un-flushed steps: flush first", layer,
                        c->pending_count[layer]);
    }
    c->pending_last[layer] = (int)c->pending.size();
    ++c->pending_count[layer];
    c->pending.push_back({layer, q, k, *d});
    return 0;
}

int daam_tap_qk_enqueue_many(DaamCtx* c, int n, const int32_t* layers, const void* const* q, const void* const* k,
                             const DaamQKDesc* const* descs)
{
    if (!c || n < 0 || (n > 0 && (!layers || !q || !k || !descs))) return fail(DAAM_E_INVALID, "NULL argument");
    const size_t before = c->pending.size();
    for (int i = 0; i < n; ++i) {
        int rc = daam_tap_qk_enqueue(c, layers[i], q[i], k[i], descs[i]);
        if (rc) {
            // all or nothing: rebuild the per-layer bookkeeping for the surviving prefix
            std::vector<Pending> keep(c->pending.begin(), c->pending.begin() + before);
            c->drop_pending();
            for (auto& p : keep) (void)daam_tap_qk_enqueue(c, p.layer, p.q, p.k, &p.d);
            return rc;
        }
    }
    return 0;
}

int daam_tap_pending(DaamCtx* c, int* n_calls, int* max_steps)
{
    if (!c) return fail(DAAM_E_INVALID, "ctx is NULL");
    std::vector<int> cnt(c->max_layers, 0);
    int mx = 0;
    for (auto& p : c->pending) mx = std::max(mx, ++cnt[p.layer]);
    if (n_calls) *n_calls = (int)c->pending.size();
    if (max_steps) *max_steps = mx;
    return 0;
}

int daam_tap_flush(DaamCtx* c, void* stream)
{
    if (!c) return fail(DAAM_E_INVALID, "ctx is NULL");
    if (c->pending.empty()) return 0;
    DeviceGuard on_device(c);
    hipStream_t s = (hipStream_t)stream;
    const int in_dtype = c->pending.front().d.in_dtype;
    // group the recorded calls by layer (first-seen order, steps in recorded order), and the
    // layers by kernel: MFMA k-step count ceil(d/16), or 0 = generic kernel.
    std::vector<int> order;
    std::vector<int> slot(c->max_layers, -1);
    std::vector<std::vector<const Pending*>> per;
    std::vector<int> kind;
    for (auto& p : c->pending) {
        if (slot[p.layer] < 0) {
            slot[p.layer] = (int)order.size();
            order.push_back(p.layer);
            per.emplace_back();
            kind.push_back(tap_mfma_ksteps(p.d.head_dim));
        }
        const int i = slot[p.layer];
        if (!use_mfma(c, p.d, p.q, p.k)) kind[i] = 0;
        else if (kind[i] != 0) {
            // every step of the layer must qualify for the specialised kernel, else the generic MFMA one
            const int want = mfma_kind(c, p.d, p.q, p.k);
            if (per[i].empty()) kind[i] = want;
            else if (kind[i] != want) kind[i] = p.d.in_dtype == DAAM_BF16 ? 0 : tap_mfma_ksteps(p.d.head_dim);
        }
        per[i].push_back(&p);
    }
    // A launch that mixes head dims (SD-v1.5: 40 / 80 / 160 = the head_dim-64 kernel and the two wide ones side by side, whose LDS
    // footprints keep them from sharing CUs): every such layer on the chunked kernel instead -- ONE kind of workgroup, one launch,
    // no side streams, bit-identical sums (tests/test_gpu_chunked.py; +11 ... 13 % heat maps / s on the SD-v1.5 workload).
    if (c->tap_chunked == 2) {
        bool k65 = false, k67 = false, k69 = false;
        for (int kd : kind) {
            k65 = k65 || kd == 65;
            k67 = k67 || kd == 67;
            k69 = k69 || kd == 69;
        }
        bool k66 = false, k70 = false;                         // bf16 pipelines: head_dim <= 64 -> 66, wider heads -> 70 (use_chunk)
        for (int kd : kind) {
            k66 = k66 || kd == 66;
            k70 = k70 || kd == 70;
        }
        if ((int)k65 + (int)k67 + (int)k69 >= 2 || (k66 && k70)) {   // (an SDXL launch -- one kind -- never gets here: no per-call checks)
            bool all_ok = true;
            for (size_t i = 0; i < kind.size() && all_ok; ++i)
                if (kind[i] == 65 || kind[i] == 66 || kind[i] == 67 || kind[i] == 69)
                    for (const Pending* p : per[i]) all_ok = all_ok && chunk_ok(c, p->d, p->q, p->k);
            if (all_ok)
                for (int& kd : kind)
                    if (kd == 65 || kd == 66 || kd == 67 || kd == 69) kd = 70;
        }
    }
    // fp16 layers of head_dim 40 / 80 / 160 (SD-v1.x) whose every recorded step qualifies: the slab kernel -- whole 128-byte lines of Q,
    // ONE launch for the three head dims (bit-identical sums: tests/test_gpu_slab.py).  DAAM_TAP_CHUNKED=0 / 1 pin the older kernels.
    if (c->tap_slab && c->tap_chunked == 2 && in_dtype == DAAM_F16) {
        for (size_t i = 0; i < kind.size(); ++i) {
            if (!(kind[i] == 65 || kind[i] == 67 || kind[i] == 69 || kind[i] == 70) || !tap_slab_heads(per[i][0]->d.head_dim)) continue;
            bool ok = true;
            for (const Pending* p : per[i]) ok = ok && slab_ok(c, p->d, p->q, p->k);
            if (ok) kind[i] = 71;
        }
    }
    std::vector<int> kinds;
    for (int kd : kind)
        if (std::find(kinds.begin(), kinds.end(), kd) == kinds.end()) kinds.push_back(kd);
    int rc = 0;
    int grid_total = 0;
    // pass 1: the tables of every kernel kind -> ring -> device (all on the caller's stream)
    struct Prepared { int kd; TapLaunch L; int max_d; int min_d; int all_round; size_t ring_begin, ring_end; bool w8; };
    std::vector<Prepared> prepared;
    for (int kd : kinds) {
        size_t n_layers = 0, n_ptrs = 0;
        for (size_t i = 0; i < order.size(); ++i)
            if (kind[i] == kd) { ++n_layers; n_ptrs += per[i].size(); }
        int tile = kd == 71 ? tap_slab_tile_pixels() : kd ? tap_mfma_tile_pixels() : kTapPixels;
        bool w8 = false;
        if ((kd == 65 || kd == 66) && !c->no_w8) {
            // head_dim-64 launches with fp16 Q / K and fp16 sums: 256-pixel tiles on eight-wave workgroups (one K tile for twice the pixels)
            bool full64 = true;
            for (size_t i = 0; i < order.size(); ++i)
                if (kind[i] == kd) full64 = full64 && per[i][0]->d.head_dim == 64;
            const int t8 = tap_d64_tile_pixels(in_dtype, c->acc_dtype, full64 ? 1 : 0);
            w8 = t8 != tile;
            tile = t8;
        }
        if (!kd) {                                           // the generic kernel reads the sums first
            for (size_t i = 0; i < order.size() && !rc; ++i)
                if (kind[i] == kd) rc = ensure_zeroed(c->layers[order[i]], s);
            if (rc) break;
        }
        // table entries of this kind: one per layer -- except that the slab kernel (71) may take the LAST pixels of a head_dim-40 layer as a second
        // entry with 16-pixel tiles (see below)
        struct Ent { size_t i; int rank, px_begin, px_end, tile; };
        std::vector<Ent> ents;
        // slab kernel: the layers segment by segment (one cost per workgroup each): head_dim 160 first (few, light workgroups with the longest
        // step chains), 40 (the bulk), 80 (short chains), and last the TAIL of the head_dim-40 layers in half-size workgroups: 2.2 rounds of
        // indivisible 50-step chains leave a third of the chip idle for the last 100 us of the launch; half-length units empty it more evenly
        // (DAAM_SLAB_TAIL = percent of a head_dim-40 layer's pixels that go there; bit-identical sums either way).  Every XCD takes an eighth of
        // each segment.  Round 6 searched the order with a list-scheduling model (tools/slab_order_model.py) and measured its best candidate
        // ("columns": first halves heavy-first, second halves light-first, half-size units last) on the chip: 3 % SLOWER than this order for
        // every tail share (LABNOTES R6.2) -- a chain's speed depends on what shares its CU, which the model does not know.
        auto seg_rank = [](int d) { return d == 160 ? 0 : d == 40 ? 1 : 2; };
        for (size_t i = 0; i < order.size(); ++i) {
            if (kind[i] != kd) continue;
            const DaamQKDesc& d0 = per[i][0]->d;
            if (kd != 71) { ents.push_back({i, 0, 0, d0.hw, tile}); continue; }
            const int r = seg_rank(d0.head_dim);
            const int tail_px = (r == 1 && d0.hw >= 64) ? (int)((int64_t)d0.hw * c->slab_tail_pct / 100 / 32) * 32 : 0;
            if (d0.hw - tail_px > 0) ents.push_back({i, r, 0, d0.hw - tail_px, tile});
            if (tail_px > 0) ents.push_back({i, 3, d0.hw - tail_px, d0.hw, tile / 2});
        }
        if (kd == 71) std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.rank < b.rank; });
        n_layers = ents.size();
        const size_t bytes_layers = n_layers * sizeof(TapLayer), bytes = bytes_layers + n_ptrs * sizeof(TapPtr);
        size_t off = 0;
        hipError_t e = c->ring.alloc(bytes, &off);
        if (e != hipSuccess) { rc = fail((int)e, "upload ring: %s", hipGetErrorString(e)); break; }
        TapLayer* hl = reinterpret_cast<TapLayer*>(c->ring.host + off);
        TapPtr* hp = reinterpret_cast<TapPtr*>(c->ring.host + off + bytes_layers);
        int wg = 0, ptr = 0, max_d = 0, min_d = 1 << 30, all_round = 1;
        std::vector<int> ptr_of(order.size(), -1);             // a layer's step pointers are written once, both of its entries point at them
        for (size_t i = 0; i < order.size(); ++i) {
            if (kind[i] != kd) continue;
            ptr_of[i] = ptr;
            for (auto* p : per[i]) { hp[ptr].q = p->q; hp[ptr].k = p->k; ++ptr; }
        }
        size_t j = 0;
        int seg_begin[kMaxSlabSegs + 1] = {0}, n_seg = 0;
        int last_rank = -1;
        for (const Ent& en : ents) {
            const auto& v = per[en.i];
            fill_layer(c, c->layers[order[en.i]], v[0]->d, en.tile, &hl[j]);
            hl[j].wg_begin = wg;
            hl[j].n_steps = (int)v.size();
            hl[j].ptr_begin = ptr_of[en.i];
            hl[j].px_begin = en.px_begin;
            hl[j].px_end = en.px_end;
            hl[j].tile_px = en.tile;
            hl[j].tiles_per_head = (en.px_end - en.px_begin + en.tile - 1) / en.tile;
            if (kd == 71) {
                if (en.rank != last_rank) { seg_begin[n_seg++] = wg; last_rank = en.rank; }
                wg += hl[j].heads_kept / tap_slab_heads(v[0]->d.head_dim) * hl[j].tiles_per_head;   // tiles_per_head = tiles per slab
            } else {
                wg += hl[j].heads_kept * hl[j].tiles_per_head;
            }
            max_d = std::max(max_d, v[0]->d.head_dim);
            min_d = std::min(min_d, v[0]->d.head_dim);
            all_round = all_round && v[0]->d.round_logits;
            ++j;
        }
        seg_begin[n_seg] = wg;
        // daam_finalize_prepare: the output of the finalize that follows this launch is cleared by this (first) upload kernel
        const bool fold = c->fold_out && c->fold_stream == s;
        e = c->ring.commit(off, bytes, s, fold ? c->fold_out : nullptr, fold ? c->fold_bytes : 0);
        if (e != hipSuccess) { rc = fail((int)e, "table upload: %s", hipGetErrorString(e)); break; }
        if (fold) {
            c->prep_out = c->fold_out;
            c->prep_stream = s;
            c->fold_out = nullptr;
        }
        Prepared pr;
        pr.kd = kd;
        memset(&pr.L, 0, sizeof pr.L);
        pr.L.layers = reinterpret_cast<const TapLayer*>(c->ring.dev + off);
        pr.L.ptrs = reinterpret_cast<const TapPtr*>(c->ring.dev + off + bytes_layers);
        pr.L.n_layers = (int)n_layers;
        pr.L.tokens = c->tokens;
        pr.L.total_wgs = wg;
        pr.L.wgs_per_xcd = (wg + 7) / 8;
        pr.L.n_seg = n_seg;
        for (int k = 0; k <= kMaxSlabSegs; ++k) pr.L.seg_begin[k] = seg_begin[k];
        pr.max_d = max_d;
        pr.min_d = min_d;
        pr.all_round = all_round;
        pr.w8 = w8;
        pr.ring_begin = c->ring.cur_begin;
        pr.ring_end = c->ring.cur_end;
        prepared.push_back(pr);
    }
    // pass 2: one launch per kind.  A flush with several kinds (SD-v1.5: head_dim 40 / 80 / 160) has kernels of a
    // few dozen to a few hundred workgroups x 50 sequential steps each, which leave most of the chip idle when run
    // one after the other.  The largest stays on the caller's stream, every other kind gets its own auxiliary
    // non-blocking stream, forked from / joined to the caller's stream by events (all tables are uploaded before
    // the fork) and launched FIRST so that its few workgroups are resident when the large grid fills the rest.
    const bool side = !rc && prepared.size() > 1 && prepared.size() <= (size_t)DaamCtx::kAux + 1 && !c->no_side_stream;
    if (side) {
        hipError_t ae = ensure_aux(c);
        if (ae != hipSuccess) rc = fail((int)ae, "auxiliary streams: %s", hipGetErrorString(ae));
    }
    size_t main_idx = 0;
    for (size_t i = 1; i < prepared.size(); ++i)
        if (prepared[i].L.total_wgs > prepared[main_idx].L.total_wgs) main_idx = i;
    const bool ev_started = c->profile && !rc && !prepared.empty();
    if (ev_started) (void)hipEventRecord(c->prof_event(0, 0), s);
    bool forked = false;
    if (side && !rc) {
        if (hipEventRecord(c->aux_fork, s) != hipSuccess) rc = fail(DAAM_E_STATE, "stream fork failed");
        else forked = true;
    }
    std::vector<size_t> launch_order;                        // side kinds first, the main one last
    for (size_t i = 0; i < prepared.size(); ++i)
        if (!forked || i != main_idx) launch_order.push_back(i);
    if (forked) launch_order.push_back(main_idx);
    int n_side = 0;
    // start gate: the side kernels' workgroups count themselves in, the main kernel waits (one wave, bounded) until they are
    // resident -- only for the kernels that carry the counter (the MFMA kinds).
    // A gate that runs into its 200 us timeout means the side kernels were NOT running beside the caller's stream at that moment (one
    // hardware queue, GPU_MAX_HW_QUEUES; or another process held the GPU): every gated flush then pays the 200 us for nothing.  The
    // timeout counter is pinned host memory the gate kernels bump; the host reads it here WITHOUT synchronising, so it lags the
    // enqueued flushes by however many are still in flight.  Rule (robust against that lag): since the gate was last armed, at least
    // three timeouts AND at least half of the gated flushes enqueued so far timed out -> the gate rests for 64 flushes (said once),
    // then is armed again with fresh counts.
    if (forked && c->gate_timeouts) {
        const unsigned now = *reinterpret_cast<volatile unsigned*>(c->gate_timeouts);
        if (c->gate_off_until && c->n_flushes >= c->gate_off_until) { c->gate_off_until = 0; c->gate_timeouts_seen = now; c->gate_enqueued = 0; }
        if (!c->gate_off_until) {
            const unsigned timeouts = now - c->gate_timeouts_seen;         // since armed (unsigned wrap-around is fine)
            if (timeouts >= 3 && 2 * (unsigned long long)timeouts >= c->gate_enqueued) {
                c->gate_off_until = c->n_flushes + 64;
                if (!c->gate_said) {
                    c->gate_said = true;
                    fprintf(stderr, "libdaam_hip: the start gate of %u of %llu multi-kernel tap launches timed out (side streams not concurrent with "
                                    "the caller's stream); the gate rests for 64 launches\n", timeouts, c->gate_enqueued);
                }
            }
        }
    }
    bool gate = forked && !c->no_start_gate && !c->gate_off_until && c->d_started;
    for (size_t i = 0; i < prepared.size(); ++i)
        if (i != main_idx && !prepared[i].kd) gate = false;
    unsigned gate_wgs = 0;
    std::string launched_names;
    for (size_t pi : launch_order) {
        if (rc) break;
        Prepared& pr = prepared[pi];
        hipStream_t ks = s;
        if (forked && pi != main_idx) {
            ks = c->aux_stream[n_side];
            if (hipStreamWaitEvent(ks, c->aux_fork, 0) != hipSuccess) { rc = fail(DAAM_E_STATE, "stream fork failed"); break; }
            if (gate) { pr.L.started = c->d_started; gate_wgs += (unsigned)pr.L.total_wgs; }
        } else if (gate && gate_wgs) {
            c->started_target += gate_wgs;                     // unsigned wrap-around is fine: the kernel compares differences
            hipError_t ge = launch_start_gate(c->d_started, c->started_target, 200, c->gate_timeouts_dev, s);
            if (ge != hipSuccess) { rc = fail((int)ge, "start gate: %s", hipGetErrorString(ge)); break; }
        }
        int grid = 0;
        hipError_t e = (pr.kd == 65 || pr.kd == 66) ? launch_tap_d64(pr.L, in_dtype, c->acc_dtype, c->fast_exp && pr.all_round, pr.min_d == 64 && pr.max_d == 64, pr.w8 ? 1 : 0, ks, &grid, &c->last_lds[0])
                     : (pr.kd == 67 || pr.kd == 69) ? launch_tap_wide(pr.L, c->acc_dtype, pr.max_d, c->fast_exp && pr.all_round, ks, &grid, &c->last_lds[0])
                     : pr.kd == 70 ? launch_tap_chunk(pr.L, in_dtype, c->acc_dtype, c->fast_exp && pr.all_round, pr.min_d != pr.max_d, ks, &grid, &c->last_lds[0])
                     : pr.kd == 71 ? launch_tap_slab(pr.L, c->acc_dtype, c->fast_exp && pr.all_round, ks, &grid, &c->last_lds[0])
                     : pr.kd ? launch_tap_mfma(pr.L, c->acc_dtype, pr.max_d, c->fast_exp && pr.all_round, ks, &grid, &c->last_lds[0])
                             : launch_tap_generic(pr.L, in_dtype, c->acc_dtype, pr.max_d, ks, &grid, &c->last_lds[0]);
        grid_total += grid;
        if (e != hipSuccess) { rc = fail((int)e, "tap launch: %s", hipGetErrorString(e)); break; }
        launched_names += (launched_names.empty() ? "" : "+") + std::string(tap_kernel_name(pr.kd));
        e = c->ring.release_range(pr.ring_begin, pr.ring_end, ks);
        if (e != hipSuccess) { rc = fail((int)e, "event record: %s", hipGetErrorString(e)); break; }
        if (ks != s) {
            if (hipEventRecord(c->aux_join[n_side], ks) != hipSuccess) { rc = fail(DAAM_E_STATE, "stream join failed"); break; }
            ++n_side;
        }
        for (size_t i = 0; i < order.size(); ++i)
            if (kind[i] == pr.kd) { c->layers[order[i]].dirty = true; c->layers[order[i]].zero_pending = false; }
    }
    // a flush that failed part-way may have announced side workgroups that never started: counter and target would disagree for
    // good (every later gate a silent no-op or a full timeout), so the context stops gating
    if (rc && gate) c->no_start_gate = 1;
    if (forked && gate && gate_wgs != 0) ++c->gate_enqueued;
    // join (after the main kernel is enqueued): the caller's stream continues when every side kernel is done
    for (int i = 0; i < n_side; ++i)
        if (hipStreamWaitEvent(s, c->aux_join[i], 0) != hipSuccess) rc = rc ? rc : fail(DAAM_E_STATE, "stream join failed");
    if (c->profile && ev_started) { (void)hipEventRecord(c->prof_event(0, 1), s); ++c->hist_count[0]; }
    c->last_grid[0] = grid_total;
    c->last_block[0] = 256;
    for (auto& pr : prepared)
        if (pr.w8 || pr.kd == 71) c->last_block[0] = 512;
    c->last_kernels[0] = launched_names;
    c->last_flush_kernels = (int)launch_order.size();
    c->last_flush_side = n_side;
    c->last_flush_steps = 0;
    for (auto& v : per) c->last_flush_steps = std::max(c->last_flush_steps, (int)v.size());
    ++c->n_flushes;
    c->drop_pending();
    c->fold_out = nullptr;                                     // one-shot: never carried to a later launch
    return rc;
}

int daam_tap_probs(DaamCtx* c, int layer, const void* probs, int in_dtype, int batch_heads, int hw, int tokens,
                   void* stream)
{
    if (!c || !probs) return fail(DAAM_E_INVALID, "NULL argument");
    if (layer < 0 || layer >= c->max_layers || !c->layers[layer].configured)
        return fail(DAAM_E_STATE, "layer %d not configured", layer);
    if (!c->pending.empty()) return fail(DAAM_E_STATE, "probs tap with deferred taps pending: flush first");
    const Layer& l = c->layers[layer];
    if (in_dtype != DAAM_F16 && in_dtype != DAAM_F32 && in_dtype != DAAM_BF16) return fail(DAAM_E_INVALID, "in_dtype %d", in_dtype);
    if (!dtypes_compatible(in_dtype, c->acc_dtype))
        return fail(DAAM_E_INVALID, "probabilities of dtype %d cannot feed running sums of dtype %d (own dtype or f32)", in_dtype, c->acc_dtype);
    if (tokens != c->tokens) return fail(DAAM_E_INVALID, "tokens %d != context size %d", tokens, c->tokens);
    if (batch_heads - batch_heads / 2 != l.heads || hw != l.hw)
        return fail(DAAM_E_INVALID, "layer %d is [%d heads, %d positions], call has [%d kept, %d]", layer, l.heads, l.hw,
                    batch_heads - batch_heads / 2, hw);
    DeviceGuard on_device(c);
    {
        int zrc = ensure_zeroed(c->layers[layer], (hipStream_t)stream);
        if (zrc) return zrc;
    }
    ProbsLaunch L;
    L.acc = l.acc;
    L.probs = probs;
    L.heads_kept = l.heads;
    L.bh_first = batch_heads / 2;
    L.hw = hw;
    L.tokens = tokens;
    L.tiles_per_head = (hw + kTapPixels - 1) / kTapPixels;
    L.total_wgs = L.heads_kept * L.tiles_per_head;
    L.wgs_per_xcd = (L.total_wgs + 7) / 8;
    c->last_block[0] = 256;
    hipError_t e = launch_tap_probs(L, in_dtype, c->acc_dtype, (hipStream_t)stream, &c->last_grid[0], &c->last_lds[0]);
    if (e != hipSuccess) return fail((int)e, "probs tap launch: %s", hipGetErrorString(e));
    c->layers[layer].dirty = true;
    return 0;
}

int daam_key_offset(DaamCtx* c, int layer, int* offset, int* total)
{
    if (!c) return fail(DAAM_E_INVALID, "ctx is NULL");
    int off = 0, tot = 0;
    for (int i = 0; i < c->max_layers; ++i) {
        if (i == layer) off = tot;
        if (c->layers[i].configured) tot += c->layers[i].heads;
    }
    if (offset) *offset = off;
    if (total) *total = tot;
    return 0;
}

// ---- finalize: host-side plan (which keys, in which class, chunking of the pipelined x2 kernel) and the device tables it needs
namespace {
constexpr int kFinClasses = 5;       // 0 = same size (clamp + mean), 1 = x2 (32 -> 64), 2 = x4 (16 -> 64), 3 = general kernel, 4 = x0.5 (128 -> 64)
struct FinPlan {
    std::vector<FinKey> keys[kFinClasses];
    int total = 0, max_side = 0;
    bool mfma_up = false, pipe_up = false, fold_same = false;
    int pipe_chunks = 0, pipe_nk = 0, pipe_stride = 0, same_per = 0;
    size_t key_bytes = 0, ptr_bytes = 0;
    std::vector<char> tab;           // FinKey array (class order) | pointer table of the pipelined x2 kernel | folded same-size pointers
};
int fin_env(const char* name)
{
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
}
}  // namespace

// token rows a finalize call covers (ABI v6: the caller may pass the prompt's n_tokens + 2, daam/trace.py:127)
static int fin_rows(const DaamCtx* c, int n_rows) { return (n_rows <= 0 || n_rows > c->tokens) ? c->tokens : n_rows; }

static int fin_plan(DaamCtx* c, const uint8_t* key_mask, int rows, FinPlan& P)
{
    static const int env_chunks = fin_env("DAAM_FIN_CHUNKS"), env_pipe_chunks = fin_env("DAAM_FIN_PIPE_CHUNKS");   // pipelined x2 kernel only (A/B)
    auto& keys = P.keys;
    int pos = 0;
    for (int i = 0; i < c->max_layers; ++i) {
        const Layer& l = c->layers[i];
        if (!l.configured) continue;
        for (int h = 0; h < l.heads; ++h, ++pos) {
            if (key_mask && !key_mask[pos]) continue;
            FinKey k;
            k.base = static_cast<const char*>(l.acc) + (size_t)h * c->tokens * l.hw * acc_elem(c->acc_dtype);
            k.side = l.side;
            k.tab = l.tab;
            int cls = 3;
            if (!c->force_generic) {
                if (l.tab < 0 && (l.hw % 8) == 0) cls = 0;
                else if (l.tab >= 0 && finalize_up_supported(l.side, c->out_side)) cls = l.side == 32 ? 1 : 2;
                else if (l.tab >= 0 && finalize_down2_supported(l.side, c->out_side)) cls = 4;
            }
            if (cls == 3 && l.tab >= 0) P.max_side = std::max(P.max_side, l.side);
            keys[cls].push_back(k);
            ++P.total;
        }
    }
    if (P.total == 0) return fail(DAAM_E_NOMAPS, "no heat maps selected");
    if (P.max_side > 128) return fail(DAAM_E_UNSUPPORTED, "map side %d > 128 not supported by finalize", P.max_side);
    // x2 class on the matrix cores (fp16 planes, fp16-exact tap matrix)?
    P.mfma_up = !keys[1].empty() && keys[1][0].tab == c->up32_tab && c->d_up32_ops &&
                c->tab_fp16_exact[keys[1][0].tab] && !c->no_mfma_finalize;
    // ... on the software-pipelined kernel (daam_finalize_pipe.hip): workgroup = (token, key chunk), every wave walks ALL keys
    // of its chunk from a pointer table padded with the all-zero plane to one even length >= 4 (+ what the ring prefetches
    // past the end).  ~1000 workgroups of 2 waves = one resident round at 2 waves per SIMD.
    P.pipe_up = P.mfma_up && !c->no_pipe_finalize && c->d_zero_planes;
    // bf16 / f32 sums (round 6): the pipelined kernel only (bf16: the tap matrix must split into two bf16 MFMA operands); the round-2 MFMA
    // kernels behind DAAM_NO_PIPE_FINALIZE take fp16 planes
    if (c->acc_dtype == DAAM_BF16) P.pipe_up = P.pipe_up && c->d_up32_ops_bf16;
    if (c->acc_dtype != DAAM_F16) P.mfma_up = P.pipe_up;
    if (P.pipe_up) {
        const int n = (int)keys[1].size();
        const int want = env_pipe_chunks ? env_pipe_chunks : env_chunks ? env_chunks : std::max(1, (1024 + rows / 2) / rows);
        P.pipe_chunks = std::max(1, std::min(want, (n + 7) / 8));
        const int per = (n + P.pipe_chunks - 1) / P.pipe_chunks;
        P.pipe_nk = std::max(4, (per + 1) & ~1);
        P.pipe_stride = (P.pipe_nk + finalize_pipe_ring(c->acc_dtype) + 2) & ~1;
    }
    // The same-size (64 x 64) keys ride along in the pipelined kernel (every wave adds its share of them to its accumulators
    // before the x2 loop) unless they outnumber the x2 keys 2 : 1 -- then they keep their own streaming kernel.
    P.fold_same = P.pipe_up && !keys[0].empty() && c->out_side == 64 && keys[0].size() <= 2 * keys[1].size() && !c->no_fold_same;
    P.same_per = P.fold_same ? ((int)keys[0].size() + P.pipe_chunks - 1) / P.pipe_chunks : 0;
    P.key_bytes = ((size_t)P.total * sizeof(FinKey) + 63) & ~size_t(63);
    P.ptr_bytes = (size_t)P.pipe_chunks * P.pipe_stride * sizeof(unsigned long long);
    P.tab.assign(P.key_bytes + P.ptr_bytes + (size_t)P.pipe_chunks * P.same_per * sizeof(unsigned long long), 0);
    FinKey* dst = reinterpret_cast<FinKey*>(P.tab.data());
    for (auto& v : keys) {
        if (!v.empty()) memcpy(dst, v.data(), v.size() * sizeof(FinKey));
        dst += v.size();
    }
    if (P.pipe_up) {
        unsigned long long* pt = reinterpret_cast<unsigned long long*>(P.tab.data() + P.key_bytes);
        const unsigned long long zero = reinterpret_cast<unsigned long long>(c->d_zero_planes);
        const int n = (int)keys[1].size(), per = (n + P.pipe_chunks - 1) / P.pipe_chunks;
        for (int ch = 0; ch < P.pipe_chunks; ++ch)
            for (int j = 0; j < P.pipe_stride; ++j) {
                const int k = ch * per + j;
                pt[(size_t)ch * P.pipe_stride + j] = (j < per && k < n) ? reinterpret_cast<unsigned long long>(keys[1][k].base) : zero;
            }
        unsigned long long* st = pt + (size_t)P.pipe_chunks * P.pipe_stride;
        for (int ch = 0; ch < P.pipe_chunks; ++ch)
            for (int j = 0; j < P.same_per; ++j) {
                const size_t k = (size_t)ch * P.same_per + j;
                st[(size_t)ch * P.same_per + j] = k < keys[0].size() ? reinterpret_cast<unsigned long long>(keys[0][k].base) : 0ull;
            }
    }
    return 0;
}

// the device copy of the tables is the one these bytes were uploaded to, on this stream?
static bool fin_cache_hit(const DaamCtx* c, const FinPlan& P, hipStream_t s)
{
    return !c->no_fin_cache && c->fin_tab_valid && c->fin_tab_stream == s && P.tab.size() == c->fin_tab_host.size() &&
           memcmp(P.tab.data(), c->fin_tab_host.data(), P.tab.size()) == 0;
}

// may this call's tables replace the cached ones?  (not while kernels enqueued on ANOTHER stream may still be reading them)
static bool fin_cacheable(const DaamCtx* c, const FinPlan& P, hipStream_t s)
{
    return !c->no_fin_cache && c->d_fin_tab && P.tab.size() <= DaamCtx::kFinTabCap && (!c->fin_tab_valid || c->fin_tab_stream == s);
}

// tables -> pinned ring -> d_fin_tab by the upload kernel (which also clears `zero`), in stream order
static int fin_cache_upload(DaamCtx* c, const FinPlan& P, hipStream_t s, void* zero, size_t zero_bytes)
{
    size_t off = 0;
    HIP_TRY(c->ring.alloc(P.tab.size(), &off));
    memcpy(c->ring.host + off, P.tab.data(), P.tab.size());
    c->fin_tab_valid = false;
    hipError_t e = launch_upload(c->d_fin_tab, c->ring.host_dev + off, P.tab.size(), zero, zero_bytes, s);
    (void)c->ring.release(s);                                  // the staging region is free once the upload kernel has run
    if (e != hipSuccess) return fail((int)e, "table upload: %s", hipGetErrorString(e));
    c->fin_tab_host = P.tab;
    c->fin_tab_valid = true;
    c->fin_tab_stream = s;
    return 0;
}

static bool fin_out_zeroable(const DaamCtx* c, const float* out, int rows)
{
    const size_t out_bytes = sizeof(float) * rows * (size_t)c->out_side * c->out_side;
    return out_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
}

int daam_finalize_prepare(DaamCtx* c, const uint8_t* key_mask, int n_rows, float* out, void* stream)
{
    if (!c || !out) return fail(DAAM_E_INVALID, "NULL argument");
    DeviceGuard on_device(c);
    hipStream_t s = (hipStream_t)stream;
    const int rows = fin_rows(c, n_rows);
    c->prep_out = c->fold_out = nullptr;
    c->prep_rows = rows;
    if (c->no_fin_cache || !fin_out_zeroable(c, out, rows)) return 0;  // daam_finalize does everything itself
    FinPlan P;
    if (fin_plan(c, key_mask, rows, P)) return 0;                // nothing selected / unsupported: daam_finalize reports it
    const size_t out_bytes = sizeof(float) * rows * (size_t)c->out_side * c->out_side;
    if (!fin_cache_hit(c, P, s)) {
        if (!fin_cacheable(c, P, s)) return 0;
        int rc = fin_cache_upload(c, P, s, out, out_bytes);      // first call of a geometry / selection: tables + zeroing now
        if (rc) return rc;
        c->prep_out = out;
        c->prep_stream = s;
        return 0;
    }
    if (!c->pending.empty()) {                                   // the table-upload kernel of the coming tap launch clears `out`
        c->fold_out = out;
        c->fold_bytes = out_bytes;
        c->fold_stream = s;
        return 0;
    }
    hipError_t e = launch_upload(nullptr, nullptr, 0, out, out_bytes, s);
    if (e != hipSuccess) return fail((int)e, "output zeroing: %s", hipGetErrorString(e));
    c->prep_out = out;
    c->prep_stream = s;
    return 0;
}

int daam_finalize(DaamCtx* c, const uint8_t* key_mask, int n_rows, float* out, void* stream)
{
    if (!c || !out) return fail(DAAM_E_INVALID, "NULL argument");
    if (!c->pending.empty()) return fail(DAAM_E_STATE, "finalize with deferred taps pending: flush first");
    DeviceGuard on_device(c);
    hipStream_t s = (hipStream_t)stream;
    const int rows = fin_rows(c, n_rows);
    // zeroed ahead of this call (daam_finalize_prepare, same buffer, same rows, same stream)?  One-shot.
    const bool prepared = c->prep_out == out && c->prep_stream == s && c->prep_rows == rows;
    c->prep_out = c->fold_out = nullptr;
    for (auto& l : c->layers)
        if (l.configured) {
            int zrc = ensure_zeroed(l, s);
            if (zrc) return zrc;
        }
    constexpr int kClasses = kFinClasses;
    FinPlan P;
    {
        int prc = fin_plan(c, key_mask, rows, P);
        if (prc) return prc;
    }
    auto& keys = P.keys;
    const int total = P.total, max_side = P.max_side;
    const bool mfma_up = P.mfma_up, pipe_up = P.pipe_up, fold_same = P.fold_same;
    const int pipe_chunks = P.pipe_chunks, pipe_nk = P.pipe_nk, pipe_stride = P.pipe_stride, same_per = P.same_per;
    const size_t key_bytes = P.key_bytes, ptr_bytes = P.ptr_bytes, bytes = P.tab.size();
    const size_t plane = (size_t)c->out_side * c->out_side;
    const size_t out_bytes = sizeof(float) * rows * plane;
    static const int env_chunks = fin_env("DAAM_FIN_CHUNKS");
    static const int env_up_chunks = fin_env("DAAM_FIN_UP_CHUNKS");       // LDS up kernels only (A/B)
    // the output is accumulated with atomics: it is zeroed by the table-upload launch unless daam_finalize_prepare had it done
    const bool zero_in_upload = fin_out_zeroable(c, out, rows);
    if (!zero_in_upload && !prepared) {
        hipError_t ze = hipMemsetAsync(out, 0, out_bytes, s);
        if (ze != hipSuccess) return fail((int)ze, "output memset: %s", hipGetErrorString(ze));
    }
    if (c->profile) (void)hipEventRecord(c->prof_event(1, 0), s);    // timed: what this call launches (table upload + zeroing if needed, class kernels)
    void* zero_ptr = (zero_in_upload && !prepared) ? out : nullptr;
    const size_t zero_n = zero_ptr ? out_bytes : 0;
    const char* tab_dev = nullptr;
    bool ring_held = false;                                     // the tables sit in a ring region the class kernels read
    if (fin_cache_hit(c, P, s)) {
        tab_dev = c->d_fin_tab;
        if (zero_ptr) {
            hipError_t e = launch_upload(nullptr, nullptr, 0, zero_ptr, zero_n, s);
            if (e != hipSuccess) return fail((int)e, "output zeroing: %s", hipGetErrorString(e));
        }
    } else if (fin_cacheable(c, P, s)) {
        int rc = fin_cache_upload(c, P, s, zero_ptr, zero_n);
        if (rc) return rc;
        tab_dev = c->d_fin_tab;
    } else {
        size_t off = 0;
        HIP_TRY(c->ring.alloc(bytes, &off));
        memcpy(c->ring.host + off, P.tab.data(), bytes);
        hipError_t ce = c->ring.commit(off, bytes, s, zero_ptr, zero_n);
        if (ce != hipSuccess) {
            (void)c->ring.release(s);
            return fail((int)ce, "table upload: %s", hipGetErrorString(ce));
        }
        tab_dev = c->ring.dev + off;
        ring_held = true;
    }
    auto release_tab = [&]() { if (ring_held) (void)c->ring.release(s); ring_held = false; };
    const FinKey* dev = reinterpret_cast<const FinKey*>(tab_dev);
    c->last_block[1] = 256;
    c->last_grid[1] = 0;
    c->last_lds[1] = 0;
    // The round-2 MFMA kernel (DAAM_NO_PIPE_FINALIZE=1) walks a host-built chunk table of at most kFinMaxChunks chunks x 2 key
    // lanes x 64 keys, so a larger class goes out as several launches over key sub-ranges (kFinMfmaKeysPerLaunch each) -- the
    // pipelined kernel, the LDS kernel and the other classes take any key count.
    constexpr int kFinMfmaKeysPerLaunch = kFinMaxChunks * 128;
    // x2 class chunking: ~1000 workgroups (one full round at 4 workgroups per CU) measured best -- fewer leaves a ragged
    // tail, more pays the per-workgroup reduction + atomics too often; every key lane of a chunk takes at most 64 keys
    const int want_up = env_up_chunks ? env_up_chunks : env_chunks ? env_chunks : std::max(1, (1024 + rows / 2) / rows);
    // (few keys: at least 4 per wave -- a workgroup ends in a four-wave LDS reduction + 4096 atomics, which one key per wave does
    // not pay for: SD-v1.5's 48 x4 keys in 3 chunks instead of 12 take 10 us less)
    auto up_chunks = [&](int n) { return std::max(std::max(1, std::min((n + 15) / 16, want_up)), (n + 127) / 128); };
    // build the launch descriptor of every non-empty class first
    FinLaunch launches[kClasses];
    bool have[kClasses] = {false, false, false, false, false};
    for (int cls = 0; cls < kClasses; ++cls) {
        const int n = (int)keys[cls].size();
        if (n == 0) continue;
        FinLaunch& L = launches[cls];
        L.keys = dev;
        dev += n;
        L.tab_idx = c->d_tab_idx;
        L.tab_w = c->d_tab_w;
        L.out = out;
        L.n_keys = n;
        L.tokens = rows;                                       // grid dimension / bound of every class kernel; a key's planes keep their [tokens] stride
        L.out_side = c->out_side;
        L.inv_n = 1.0f / (float)total;
        L.max_side = max_side;
        L.mfma_ops = (cls == 1 && mfma_up) ? c->d_up32_ops : nullptr;
        if (cls == 0) {
            // 154 workgroups per chunk: 4 chunks for the 100 same-size keys of SDXL-1024, up to 9 (1386 workgroups) for the 1000 of
            // SDXL-2048.  Round 2 had 16 there; round 4's sweep (tools/exp/fin_chunks_sweep.sh: 4 ... 64 chunks) has its optimum at
            // 8 - 10 -- the x0.5 class now streams beside this one on a side stream, and every chunk ends in 315 k atomics per
            // token plane: 0.186 -> 0.159 ms for the SDXL-2048 call (0.59 -> 0.69 of the HBM peak)
            L.n_chunks = std::max(1, std::min(n, env_chunks ? env_chunks : std::min(9, std::max(4, n / 32))));
        } else if (cls == 3) {
            L.n_chunks = std::max(1, std::min(n, 32));
        } else {
            // each wave takes keys first, first + 4*n_chunks, ...: at most 64 per wave (4 key lanes per workgroup; the MFMA
            // kernel of class 1 has 2 and is chunked per launch below)
            L.n_chunks = up_chunks(n);
        }
        memset(L.chunk_begin, 0, sizeof L.chunk_begin);
        have[cls] = true;
    }
    // the MFMA kernel's launches: key sub-ranges of at most kFinMfmaKeysPerLaunch keys, each with its own chunk table
    std::vector<FinLaunch> up_parts;
    if (mfma_up && !pipe_up) {
        const FinLaunch& U = launches[1];
        for (int begin = 0; begin < U.n_keys; begin += kFinMfmaKeysPerLaunch) {
            FinLaunch P = U;
            P.keys = U.keys + begin;
            P.n_keys = std::min(kFinMfmaKeysPerLaunch, U.n_keys - begin);
            P.n_chunks = std::min(up_chunks(P.n_keys), kFinMaxChunks);
            finalize_chunk_ranges(P.n_keys, P.n_chunks, &P);
            up_parts.push_back(P);
        }
        launches[1] = up_parts[0];                             // what the paired launch takes (single part)
    }
    // SDXL-1024 in fp16: the same-size and the x2 class side by side in ONE launch
    const bool paired = mfma_up && !pipe_up && up_parts.size() == 1 && have[0] && !c->no_paired_finalize;
    // Several classes: the issue-bound x2 kernel keeps the caller's stream; every other class (HBM streams with few
    // registers: their waves fit beside the two heavy waves of a SIMD) goes to an auxiliary stream forked from / joined to the
    // caller's by events, launched FIRST -- SDXL-1024: 63 MB of same-size planes stream under 158 MB of x2 planes; SD-v1.5:
    // three classes side by side instead of three serial launches.
    if (fold_same) have[0] = false;                            // done inside the pipelined kernel
    int n_classes = 0;
    for (int cls = 0; cls < kClasses; ++cls) n_classes += have[cls] ? 1 : 0;
    // (an event fork / join costs ~15 us of queue latency per finalize call: only worth it for a side class of tens of MB --
    // SD-v1.5's 1.6 MB x4 class runs 10 us faster serially behind the pipelined kernel)
    size_t side_bytes = 0;
    for (int cls = 0; cls < kClasses; ++cls)
        if (have[cls] && cls != 1)
            for (auto& k : keys[cls]) side_bytes += (size_t)c->tokens * k.side * k.side * acc_elem(c->acc_dtype);
    bool fork = pipe_up && n_classes > 1 && n_classes <= DaamCtx::kAux + 1 && !c->no_side_stream && side_bytes >= ((size_t)16 << 20);
    if (fork) {
        hipError_t ae = ensure_aux(c);
        if (ae != hipSuccess || hipEventRecord(c->aux_fork, s) != hipSuccess) fork = false;    // serial launches still correct
    }
    int n_side = 0;
    std::string launched_names;
    auto names = [&](const char* kernel, const char* what) {
        launched_names += (launched_names.empty() ? "" : "+") + std::string(kernel) + "<" + what + ">";
    };
    auto launch_class = [&](int cls, hipStream_t ks, int* grid, int* lds) -> hipError_t {
        const FinLaunch& L = launches[cls];
        if (cls == 1 && pipe_up) {
            FinPipeLaunch PL;
            PL.key_ptrs = reinterpret_cast<const unsigned long long*>(tab_dev + key_bytes);
            PL.same_ptrs = fold_same ? reinterpret_cast<const unsigned long long*>(tab_dev + key_bytes + ptr_bytes) : nullptr;
            PL.same_per = same_per;
            PL.mfma_ops = c->acc_dtype == DAAM_BF16 ? c->d_up32_ops_bf16 : c->d_up32_ops;
            PL.out = out;
            PL.n_chunks = pipe_chunks;
            PL.nk_pad = pipe_nk;
            PL.ptr_stride = pipe_stride;
            PL.tokens = rows;
            PL.inv_n = L.inv_n;
            names("finalize_up32_pipe_kernel", (std::string(dtype_name(c->acc_dtype)) + (fold_same ? " + same-size keys" : "")).c_str());
            return launch_finalize_up32_pipe(PL, c->acc_dtype, ks, grid);
        }
        if (cls == 1 && paired) { names("finalize_up32_same_kernel", "f16"); return launch_finalize_up32_same(L, launches[0], ks, grid); }
        if (cls == 1 && mfma_up) {
            names("finalize_up32_mfma_kernel", "f16");
            hipError_t e = hipSuccess;
            for (size_t part = 0; part < up_parts.size() && e == hipSuccess; ++part) {
                int g = 0;
                e = launch_finalize_up(up_parts[part], 32, c->acc_dtype, 1, ks, &g);
                *grid += g;
            }
            return e;
        }
        if (cls == 0) { names("finalize_same_kernel", dtype_name(c->acc_dtype)); return launch_finalize_same(L, c->acc_dtype, ks, grid); }
        if (cls == 3) { names("finalize_kernel", dtype_name(c->acc_dtype)); return launch_finalize(L, c->acc_dtype, ks, grid, lds); }
        if (cls == 4) { names("finalize_down2_kernel", dtype_name(c->acc_dtype)); return launch_finalize_down2(L, c->acc_dtype, ks, grid); }
        names(keys[cls][0].side == 32 ? "finalize_up_kernel<32>" : "finalize_up_kernel<16>", dtype_name(c->acc_dtype));
        return launch_finalize_up(L, keys[cls][0].side, c->acc_dtype, 0, ks, grid);
    };
    const int order[kClasses] = {0, 2, 3, 4, 1};                       // the x2 class last: side kernels are resident when it fills the chip
    for (int oi = 0; oi < kClasses; ++oi) {
        const int cls = order[oi];
        if (!have[cls] || (paired && cls == 0)) continue;
        hipStream_t ks = s;
        const bool on_side = fork && cls != 1;
        if (on_side) {
            ks = c->aux_stream[n_side];
            if (hipStreamWaitEvent(ks, c->aux_fork, 0) != hipSuccess) ks = s;
        }
        int grid = 0, lds = 0;
        hipError_t e = launch_class(cls, ks, &grid, &lds);
        if (e == hipSuccess && ks != s) {
            e = hipEventRecord(c->aux_join[n_side], ks);
            ++n_side;
        }
        if (e != hipSuccess) {
            for (int i = 0; i < n_side; ++i) (void)hipStreamWaitEvent(s, c->aux_join[i], 0);
            release_tab();                                 // the table region is reusable once whatever did launch has run
            return fail((int)e, "finalize launch (class %d): %s", cls, hipGetErrorString(e));
        }
        c->last_grid[1] += grid;
        c->last_lds[1] = std::max(c->last_lds[1], lds);
    }
    for (int i = 0; i < n_side; ++i)
        if (hipStreamWaitEvent(s, c->aux_join[i], 0) != hipSuccess) { release_tab(); return fail(DAAM_E_STATE, "stream join failed"); }
    c->last_fin_side = n_side;
    c->last_kernels[1] = launched_names;
    if (c->profile) { (void)hipEventRecord(c->prof_event(1, 1), s); ++c->hist_count[1]; }
    release_tab();
    return 0;
}

int daam_epilogue_normalize(float* maps, int n_rows, int side, void* stream)
{
    if (!maps || n_rows <= 0 || side <= 0) return fail(DAAM_E_INVALID, "bad argument");
    hipError_t e = launch_normalize(maps, n_rows, side * side, (hipStream_t)stream);
    if (e != hipSuccess) return fail((int)e, "normalize launch: %s", hipGetErrorString(e));
    return 0;
}

int daam_word_heat_map(const float* maps, int side, const int32_t* idx, int n_idx, float* word_map, float* out,
                       int out_h, int out_w, int absolute, float threshold, float* workspace, void* stream)
{
    if (!maps || !idx || !word_map || !workspace) return fail(DAAM_E_INVALID, "NULL argument");
    if (n_idx <= 0 || n_idx > kMaxTokens) return fail(DAAM_E_INVALID, "n_idx %d not in 1..%d", n_idx, kMaxTokens);
    if (side <= 0 || (out && (out_h <= 0 || out_w <= 0))) return fail(DAAM_E_INVALID, "bad size");
    hipError_t e = launch_word(maps, side, idx, n_idx, word_map, out, out_h, out_w, absolute, threshold, workspace,
                               (hipStream_t)stream);
    if (e != hipSuccess) return fail((int)e, "word map launch: %s", hipGetErrorString(e));
    return 0;
}

int daam_mask_overlap(const float* a, int a_h, int a_w, const float* b, int b_h, int b_w, int n_pairs, float* sums, void* stream)
{
    if (!a || !b || !sums) return fail(DAAM_E_INVALID, "NULL argument");
    if (n_pairs <= 0 || n_pairs > 65535 || a_h <= 0 || a_w <= 0 || b_h <= 0 || b_w <= 0 || (long long)b_h * b_w > (1ll << 30))
        return fail(DAAM_E_INVALID, "bad shape: %d pairs, a %dx%d, b %dx%d", n_pairs, a_h, a_w, b_h, b_w);
    if (a_h == b_h && a_w != b_w)
        return fail(DAAM_E_INVALID, "same heights but widths %d / %d differ (the reference's a * b would not broadcast)", a_w, b_w);
    hipError_t e = launch_mask_overlap(a, a_h, a_w, b, b_h, b_w, n_pairs, sums, (hipStream_t)stream);
    if (e != hipSuccess) return fail((int)e, "mask overlap launch: %s", hipGetErrorString(e));
    return 0;
}

int daam_profile_enable(DaamCtx* c, int on)
{
    if (!c) return fail(DAAM_E_INVALID, "ctx is NULL");
    DeviceGuard on_device(c);
    if (on && !c->prof_ev[0][0])
        for (auto& pair : c->prof_ev)
            for (auto& ev : pair) HIP_TRY(hipEventCreate(&ev));

    if (on == 2) {
        for (int which = 0; which < 2; ++which)
            for (int end = 0; end < 2; ++end)
                if (c->hist_ev[which][end].empty()) {
                    std::vector<hipEvent_t> ring(DaamCtx::kProfHist, nullptr);
                    hipError_t e = hipSuccess;
                    for (auto& ev : ring)
                        if ((e = hipEventCreate(&ev)) != hipSuccess) break;
                    if (e != hipSuccess) {                     // never leave a ring with null events behind: prof_event() would hand them out
                        for (auto ev : ring)
                            if (ev) (void)hipEventDestroy(ev);
                        return fail((int)e, "profile event ring: %s", hipGetErrorString(e));
                    }
                    c->hist_ev[which][end] = std::move(ring);
                }
        c->hist_count[0] = c->hist_count[1] = 0;
    }
    c->profile = on == 2 ? 2 : on ? 1 : 0;
    return 0;
}

int daam_profile_history(DaamCtx* c, int which, float* ms, int capacity, int* n)
{
    if (!c || !ms || !n || which < 0 || which > 1 || capacity < 0) return fail(DAAM_E_INVALID, "bad argument");
    if (c->hist_ev[which][0].empty()) return fail(DAAM_E_STATE, "daam_profile_enable(ctx, 2) was never called");
    DeviceGuard on_device(c);
    const long long have = std::min<long long>(c->hist_count[which], DaamCtx::kProfHist);
    const int take = (int)std::min<long long>(have, capacity);
    *n = take;
    for (int i = 0; i < take; ++i) {                       // oldest of the last `take` launches first
        const size_t slot = (size_t)((c->hist_count[which] - take + i) % DaamCtx::kProfHist);
        HIP_TRY(hipEventSynchronize(c->hist_ev[which][1][slot]));
        HIP_TRY(hipEventElapsedTime(&ms[i], c->hist_ev[which][0][slot], c->hist_ev[which][1][slot]));
    }
    return 0;
}

int daam_profile_last_ms(DaamCtx* c, int which, float* ms)
{
    if (!c || !ms || which < 0 || which > 1) return fail(DAAM_E_INVALID, "bad argument");
    if (!c->prof_ev[which][0]) return fail(DAAM_E_STATE, "profiling was never enabled");
    DeviceGuard on_device(c);
    if (c->profile == 2) {                                     // the launches record into the ring: the newest slot, not a stale prof_ev pair
        if (c->hist_count[which] <= 0 || c->hist_ev[which][0].empty()) return fail(DAAM_E_STATE, "no launch of kind %d since daam_profile_enable(ctx, 2)", which);
        const size_t slot = (size_t)((c->hist_count[which] - 1) % DaamCtx::kProfHist);
        HIP_TRY(hipEventSynchronize(c->hist_ev[which][1][slot]));
        HIP_TRY(hipEventElapsedTime(ms, c->hist_ev[which][0][slot], c->hist_ev[which][1][slot]));
        return 0;
    }
    HIP_TRY(hipEventSynchronize(c->prof_ev[which][1]));
    HIP_TRY(hipEventElapsedTime(ms, c->prof_ev[which][0], c->prof_ev[which][1]));
    return 0;
}

int daam_clock_monitor_start(DaamCtx* c, int n_samples, int period_us)
{
    if (!c || n_samples < 2 || n_samples > kClockMaxSamples || period_us < 1) return fail(DAAM_E_INVALID, "bad argument");
    DeviceGuard on_device(c);
    if (!c->clk_host) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->clk_host), 2 * kClockMaxSamples * sizeof(unsigned long long), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->clk_dev), c->clk_host, 0));
        HIP_TRY(hipStreamCreateWithFlags(&c->clk_stream, hipStreamNonBlocking));
    }
    HIP_TRY(hipStreamSynchronize(c->clk_stream));
    memset(c->clk_host, 0, 2 * kClockMaxSamples * sizeof(unsigned long long));
    c->clk_samples = n_samples;
    hipError_t e = launch_clock_monitor(c->clk_dev, n_samples, period_us, c->clk_stream);
    if (e != hipSuccess) return fail((int)e, "clock monitor launch: %s", hipGetErrorString(e));
    return 0;
}

int daam_clock_monitor_read(DaamCtx* c, float* mhz, int capacity, int* n_intervals)
{
    if (!c || !mhz || !n_intervals) return fail(DAAM_E_INVALID, "NULL argument");
    if (!c->clk_host || c->clk_samples < 2) return fail(DAAM_E_STATE, "the clock monitor was never started");
    DeviceGuard on_device(c);
    HIP_TRY(hipStreamSynchronize(c->clk_stream));
    int n = 0;
    for (int i = 1; i < c->clk_samples && n < capacity; ++i) {
        const unsigned long long* a = c->clk_host + 2 * (i - 1), *b = c->clk_host + 2 * i;
        if (b[1] > a[1] && b[0] > a[0]) mhz[n++] = (float)((double)(b[0] - a[0]) / (double)(b[1] - a[1]) * 100.0);   // reference: 100 MHz
    }
    *n_intervals = n;
    return 0;
}

int daam_last_flush(DaamCtx* c, int* n_kernels, int* n_side_streams, int* max_steps, long long* n_flushes)
{
    if (!c) return fail(DAAM_E_INVALID, "ctx is NULL");
    if (n_kernels) *n_kernels = c->last_flush_kernels;
    if (n_side_streams) *n_side_streams = c->last_flush_side;
    if (max_steps) *max_steps = c->last_flush_steps;
    if (n_flushes) *n_flushes = c->n_flushes;
    return 0;
}

int daam_last_kernels(DaamCtx* c, int which, char* names, int capacity)
{
    if (!c || !names || capacity <= 0 || which < 0 || which > 1) return fail(DAAM_E_INVALID, "bad argument");
    snprintf(names, (size_t)capacity, "%s", c->last_kernels[which].c_str());
    return 0;
}

int daam_last_launch(DaamCtx* c, int which, int* grid, int* block, int* lds_bytes)
{
    if (!c || which < 0 || which > 1) return fail(DAAM_E_INVALID, "bad argument");
    if (grid) *grid = c->last_grid[which];
    if (block) *block = c->last_block[which];
    if (lds_bytes) *lds_bytes = c->last_lds[which];
    return 0;
}

}  // extern "C"
