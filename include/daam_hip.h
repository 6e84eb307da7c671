/* daam_hip.h -- C ABI of libdaam_hip.so: the MI355X (gfx950) heat-map extraction path.
 *
 * This is the drop-in boundary for the ONE hot path of castorini/daam v0.2.0
 * (paths below are relative to the reference checkout):
 *
 *   tap       = what UNetCrossAttentionHooker.__call__ does between get_attention_scores and
 *               bmm (daam/trace.py:276-294): softmax(scale*Q K^T) -> keep the conditional
 *               half of the batch*heads dim -> [heads, tokens, h, w] -> per-(layer, head)
 *               running sum (RawHeatMapCollection.update, daam/heatmap.py:153-156).
 *   finalize  = DiffusionHeatMapHooker.compute_global_heat_map (daam/trace.py:103-130):
 *               key filter -> bicubic resize to x*x -> clamp(min=0) -> mean over keys
 *               (-> token crop is a view; optional per-pixel normalisation = epilogue).
 *   word maps = GlobalHeatMap.compute_word_heat_map + WordHeatMap.expand_as
 *               (daam/heatmap.py:121-123, 77-93)  [SURVEY.md section 8f row f1].
 *   overlap   = evaluate.compute_iou / compute_ioa (daam/evaluate.py:14-35)  [row f4].
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch types.
 *   - every `const void*` / `void*` data pointer is a DEVICE pointer owned by the caller
 *     unless the parameter is documented as "host".
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Every call only
 *     ENQUEUES work on that stream and returns; nothing synchronises the host except
 *     daam_ctx_destroy and the (rare) wrap-around of the internal upload ring.
 *   - return value: 0 = ok; > 0 = a hipError_t from the HIP runtime; < 0 = DAAM_E_* below.
 *     daam_last_error() gives the message of the last failure on the calling thread.
 *   - not thread-safe per context (the reference is not re-entrant either: it serialises
 *     generation with a lock, daam/run/demo.py:69,88).
 *   - a context belongs to the HIP device that was current at daam_ctx_create; its entry points switch to that
 *     device for the duration of the call when another one is current.
 */
#ifndef DAAM_HIP_H
#define DAAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAAM_ABI_VERSION 6

/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#define DAAM_API __attribute__((visibility("default")))

/* element types of activations (q, k, probs) and of the running sums */
#define DAAM_F16 0
#define DAAM_F32 1
#define DAAM_BF16 2         /* bfloat16 pipelines: bf16 logits / probabilities / sums (MFMA taps for every head_dim that is a multiple of 8, <= 256) */

/* DAAM_E_* */
#define DAAM_E_INVALID   (-1)   /* bad argument / shape / dtype */
#define DAAM_E_STATE     (-2)   /* call not valid in the current state (e.g. layer not configured) */
#define DAAM_E_NOMAPS    (-3)   /* finalize selected zero keys (reference: RuntimeError, trace.py:118-124) */
#define DAAM_E_UNSUPPORTED (-4)

typedef struct DaamCtx DaamCtx;

/* How one cross-attention call lays out its projected query / key.
 * Logical q is [batch, heads, hw, head_dim], k is [batch, heads, tokens, head_dim]; head_dim is
 * contiguous (stride 1), the other strides are given in ELEMENTS so that both the
 * head_to_batch_dim layout [B*H, S, d] (trace.py:272-273) and the raw to_q/to_k output
 * [B, S, H*d] (trace.py:262,269) can be tapped without a copy.
 * The reference keeps batch*heads indices [BH/2, BH) (trace.py:240): with classifier-free
 * guidance (batch 2) that is the conditional prompt, all heads. */
typedef struct DaamQKDesc {
    int32_t in_dtype;        /* DAAM_F16 | DAAM_F32 | DAAM_BF16: dtype of q and k (the pipeline dtype) */
    int32_t batch;           /* B */
    int32_t heads;           /* H */
    int32_t hw;              /* query positions = h*w, square (trace.py:233) */
    int32_t tokens;          /* key positions; only ctx->tokens (77) is tapped (trace.py:289) */
    int32_t head_dim;        /* d */
    int32_t round_logits;    /* 1: round scale*q.k to in_dtype before softmax (baddbmm output
                                dtype in diffusers get_attention_scores); 0: upcast_attention */
    float   scale;           /* attn.scale = head_dim ** -0.5 */
    int64_t q_stride_b, q_stride_h, q_stride_p;
    int64_t k_stride_b, k_stride_h, k_stride_t;
} DaamQKDesc;

/* ---- context ---------------------------------------------------------------------------
 * One context per trace object per device (reference: one RawHeatMapCollection per
 * DiffusionHeatMapHooker, trace.py:31).  tokens = context_size (77, trace.py:194);
 * out_side = int(sqrt(latent_hw)) (64, or 96 for 768-px SD-2.x; trace.py:32-33,109);
 * acc_dtype = dtype of the running sums: DAAM_F16 / DAAM_BF16 reproduce the reference's fp16 / bf16
 * sums on an fp16 / bf16 pipeline bit-for-bit in the add (heatmap.py:156), DAAM_F32 is the accuracy
 * mode.  Activations must have the dtype of the sums, or the sums must be DAAM_F32. */
DAAM_API int daam_ctx_create(int max_layers, int tokens, int out_side, int acc_dtype, DaamCtx** out);
DAAM_API int daam_ctx_destroy(DaamCtx* ctx);

/* Declare layer `layer` (= position in UNetCrossAttentionLocator.locate order, trace.py:45,50):
 * `heads` = kept batch*heads entries (BH - BH/2), `side` = sqrt(hw), `factor` =
 * int(sqrt(latent_hw // hw)) (trace.py:285).  `acc` = caller-owned zero-initialised device
 * buffer [heads, tokens, side, side] of acc_dtype, or NULL to let the library allocate it. */
DAAM_API int daam_layer_configure(DaamCtx* ctx, int layer, int heads, int side, int factor, void* acc);
DAAM_API int daam_layer_acc(DaamCtx* ctx, int layer, void** acc, size_t* bytes);
/* The caller is about to write into the layer's sums itself (RawHeatMapCollection.update called by hand,
 * heatmap.py:153-156): a zeroing still owed to the buffer since the last daam_reset is enqueued on `stream` first,
 * and the layer counts as holding sums (the next tap adds to them, the next daam_reset clears them). */
DAAM_API int daam_layer_touch(DaamCtx* ctx, int layer, void* stream);
/* Forget layer `layer`: the context drops its pointer to the sums (a library-owned buffer is freed) and never touches that
 * memory again -- not even for a zeroing still owed since daam_reset.  This is RawHeatMapCollection.clear()
 * (daam/heatmap.py:170-172) as seen from tensors the caller handed out before: the reference drops its dict and the old
 * tensors live on unchanged.  The layer must be configured again before its next tap.  DAAM_E_STATE with taps pending. */
DAAM_API int daam_layer_release(DaamCtx* ctx, int layer);

/* RawHeatMapCollection.clear (heatmap.py:170-172; called from check_inputs, trace.py:179):
 * zero every running sum and drop any un-flushed deferred taps. */
DAAM_API int daam_reset(DaamCtx* ctx, void* stream);

/* ---- tap -------------------------------------------------------------------------------
 * daam_tap_qk: one hooked cross-attention call, immediate: one kernel launch that
 * recomputes the conditional-half probabilities from q,k and adds them to the layer's sums.
 * daam_tap_qk_enqueue + daam_tap_flush: deferred form.  enqueue only records pointers (the
 * caller keeps q,k alive until the flush has been enqueued on `stream` and stream order
 * protects them); flush runs ALL recorded calls - every layer, every recorded step - as one
 * launch, adding the steps of a layer in the order they were recorded (so fp16 sums round
 * exactly like the reference's step-by-step adds) while reading and writing each running
 * sum once per flush instead of once per step.
 * daam_tap_probs: same accumulate from materialised probabilities [B*H, hw, tokens]
 * (the save_heads / load_heads path, trace.py:279-282, and any processor that already
 * holds attention_probs); bit-exact with the reference in the add. */
DAAM_API int daam_tap_qk(DaamCtx* ctx, int layer, const void* q, const void* k, const DaamQKDesc* d, void* stream);
DAAM_API int daam_tap_qk_enqueue(DaamCtx* ctx, int layer, const void* q, const void* k, const DaamQKDesc* d);
/* the same as n daam_tap_qk_enqueue calls in order (all or nothing): lets a host runtime that
 * records calls cheaply hand them over in one crossing of the FFI.  HOST arrays. */
DAAM_API int daam_tap_qk_enqueue_many(DaamCtx* ctx, int n, const int32_t* layers, const void* const* q,
                             const void* const* k, const DaamQKDesc* const* descs);
DAAM_API int daam_tap_pending(DaamCtx* ctx, int* n_calls, int* max_steps);
DAAM_API int daam_tap_flush(DaamCtx* ctx, void* stream);
DAAM_API int daam_tap_probs(DaamCtx* ctx, int layer, const void* probs, int in_dtype, int batch_heads,
                   int hw, int tokens, void* stream);

/* ---- attend: the processor's attention with the tap fused in ---------------------------------
 * One cross-attention call of the reference's processor between the projections (daam/trace.py:262-270) and the
 * output projection (:300-302):  out = softmax(scale * Q K^T) V  for every (batch, head) -- get_attention_scores
 * (daam/trace.py:276) and torch.bmm(probs, value) + batch_to_head_dim (:296-297), with their rounding points: logits and
 * probabilities rounded to the pipeline dtype, the value product accumulated in f32 and rounded once -- and, when
 * `tap` != 0, the heat-map update of the same call (the per-head loop daam/trace.py:289-294, heatmap.py:153-156) from
 * the same probabilities, in the same kernel launch: nothing [B*H, hw, 77]-sized exists, Q is read once, no Q / K
 * has to stay alive after the call.  `tap` = 0 leaves the layer's sums alone (the caller records q, k for a deferred
 * daam_tap_qk_enqueue, or the call is not tapped: reference gate daam/trace.py:289).
 * v is [batch, tokens, heads * head_dim] like k; out is written as [batch, hw, heads * head_dim] in the given strides
 * (ELEMENTS; head_dim contiguous).  Supported: DAAM_F16 and DAAM_BF16 (bf16: round_logits = 1 only), head_dim a multiple of 8 up to 160 (SDXL / SD-2.x 64, SD-v1.5 40 /
 * 80 / 160), tokens 77, hw and strides multiples of 8, 16-byte aligned pointers -- daam_attend_supported() tells; otherwise DAAM_E_UNSUPPORTED and the
 * caller uses its framework's attention plus daam_tap_qk. */
typedef struct DaamAttendDesc {
    DaamQKDesc qk;
    int64_t v_stride_b, v_stride_h, v_stride_t;
    int64_t o_stride_b, o_stride_h, o_stride_p;
} DaamAttendDesc;
DAAM_API int daam_attend_supported(const DaamAttendDesc* d, const void* q, const void* k, const void* v, const void* out);
DAAM_API int daam_attend(DaamCtx* ctx, int layer, const void* q, const void* k, const void* v, void* out,
                         const DaamAttendDesc* d, int tap, void* stream);

/* ---- finalize ---------------------------------------------------------------------------
 * compute_global_heat_map (trace.py:103-126) over the keys selected by `key_mask`:
 * HOST array, one byte per (layer, head) in layer-major order over configured layers'
 * `heads` (offset of layer l = sum of heads of layers < l, see daam_key_offset), non-zero =
 * selected.  NULL selects every key.  Writes out[n_rows, out_side, out_side] fp32 =
 * mean over selected keys of clamp(bicubic(sum_plane), 0) for the token rows [0, n_rows).
 * `n_rows` (ABI v6): the reference crops the result to the prompt's n_tokens + 2 rows
 * (trace.py:127), so a caller that knows the prompt passes that count and the planes of the
 * other tokens are neither read nor written: the rows [0, n_rows) of `out` are overwritten, the
 * rows [n_rows, tokens) -- if `out` has them at all -- are left untouched.  n_rows <= 0 or
 * > tokens means all `tokens` rows. */
DAAM_API int daam_key_offset(DaamCtx* ctx, int layer, int* offset, int* total);
DAAM_API int daam_finalize(DaamCtx* ctx, const uint8_t* key_mask, int n_rows, float* out, void* stream);
/* Optional, ABI v5 (`n_rows`: v6): announce the `key_mask` / `n_rows` / `out` / `stream` of the daam_finalize call that follows, BEFORE the deferred taps are
 * launched (between daam_tap_qk_enqueue* and daam_tap_flush), or any time before daam_finalize when nothing is pending.  The key /
 * pointer tables of the selection are kept on the device between calls (a generation's compute_global_heat_map, trace.py:103-126,
 * selects the same keys at the same addresses as the previous one) and `out` is cleared by the table-upload kernel of the tap
 * launch, so that daam_finalize itself is its class kernel(s) only.  `out` must stay allocated and untouched until that
 * daam_finalize; the announcement is one-shot and dropped by daam_reset, by a daam_finalize with other arguments (which then does
 * everything itself, as without this call) and by the next daam_tap_flush. */
DAAM_API int daam_finalize_prepare(DaamCtx* ctx, const uint8_t* key_mask, int n_rows, float* out, void* stream);

/* trace.py:129-130: maps[:n_rows] / (maps[1:n_rows-1].sum(0) + 1e-6), in place on the first
 * n_rows planes of `maps` [*, side, side] fp32. */
DAAM_API int daam_epilogue_normalize(float* maps, int n_rows, int side, void* stream);

/* ---- word maps (next row f1) -------------------------------------------------------------
 * heatmap.py:121-123 + 77-93: mean of the planes `idx[0..n_idx)` (HOST int array) of
 * maps[*, side, side] -> bicubic to out_h x out_w -> (absolute ? id : min-max normalise with
 * eps 1e-8) -> (threshold != 0 ? (x > threshold) : x).  word_map[side*side] (required) receives
 * the un-expanded mean plane; `out` [out_h, out_w] fp32 (NULL: only the mean plane is computed);
 * `workspace` >= 2 floats of device scratch for the min/max. */
DAAM_API int daam_word_heat_map(const float* maps, int side, const int32_t* idx, int n_idx, float* word_map,
                       float* out, int out_h, int out_w, int absolute, float threshold,
                       float* workspace, void* stream);

/* ---- evaluation (next row f4) --------------------------------------------------------------
 * evaluate.compute_iou / compute_ioa (daam/evaluate.py:14-35) and WordHeatMap.compute_ioa (daam/heatmap.py:95-96) for a
 * batch of n pairs: a [n, a_h, a_w] (prediction) and b [n, b_h, b_w] (truth), fp32.  When a_h != b_h -- the reference tests
 * shape[0] only -- a is resized to (b_h, b_w) with the bicubic of F.interpolate and binarised (a < 1 -> 0, else 1); then
 * sums[i] = { sum(a * b), sum(a), sum(b) }  (sums [n, 3] fp32, overwritten).  The caller forms
 * IoU = s0 / (s1 + s2 - s0 + 1e-8) and IoA = s0 / (s1 + 1e-8) in fp32. */
DAAM_API int daam_mask_overlap(const float* a, int a_h, int a_w, const float* b, int b_h, int b_w, int n_pairs, float* sums,
                               void* stream);

/* ---- misc -------------------------------------------------------------------------------- */
DAAM_API int daam_abi_version(void);
DAAM_API const char* daam_last_error(void);
/* per-kernel launch statistics of the last tap / finalize launch (for bench.py):
 * grid size and dynamic LDS bytes; 0 if nothing launched yet. */
DAAM_API int daam_last_launch(DaamCtx* ctx, int which /*0 tap,1 finalize*/, int* grid, int* block, int* lds_bytes);
/* launch structure of the last daam_tap_flush that launched something: kernels launched (one per kernel kind: SD-v1.5 has
 * head_dim 40 / 80 / 160 = three), how many of them went to auxiliary streams beside the caller's, the longest per-layer
 * step chain of the launch; `n_flushes` counts such flushes since daam_ctx_create (tests and bench.py assert the launch
 * structure a configuration is supposed to have: launches per generation, side-by-side kernels). Any pointer may be NULL. */
DAAM_API int daam_last_flush(DaamCtx* ctx, int* n_kernels, int* n_side_streams, int* max_steps, long long* n_flushes);
/* ABI v6: the names of the kernel(s) the last daam_tap_flush / daam_tap_qk (`which` 0) or daam_finalize (`which` 1) really launched,
 * '+'-separated in launch order (e.g. "tap_slab_kernel", "finalize_up32_pipe_kernel<f16>+finalize_up_kernel<16>"), copied into the HOST
 * buffer `names` (NUL-terminated, truncated to `capacity`); "" if nothing launched yet.  A report names what ran, not what the
 * environment asked for. */
DAAM_API int daam_last_kernels(DaamCtx* ctx, int which /*0 tap,1 finalize*/, char* names, int capacity);
/* kernel timing for bench.py: when enabled, every tap / finalize call brackets ITS KERNEL LAUNCHES (not
 * the table upload before them) with HIP events on the call's stream; daam_profile_last_ms waits for the
 * last pair and returns the elapsed milliseconds (the only other call that synchronises the host). */
DAAM_API int daam_profile_enable(DaamCtx* ctx, int on);
DAAM_API int daam_profile_last_ms(DaamCtx* ctx, int which /*0 tap,1 finalize*/, float* ms);
/* daam_profile_enable(ctx, 2): every launch of a kind gets its own event pair (a ring of 256), so that a whole timed region can be
 * measured without a host synchronisation inside it; daam_profile_history then returns the durations (ms, oldest first) of the last
 * min(launches since the enable, 256, capacity) launches of `which`. */
DAAM_API int daam_profile_history(DaamCtx* ctx, int which /*0 tap,1 finalize*/, float* ms, int capacity, int* n);
/* Shader-clock monitor for issue-rate rooflines: one wave on an internal stream takes `n_samples` (<= 4096) samples, `period_us`
 * apart, of the shader-cycle counter (s_memtime) and the constant 100 MHz reference counter (s_memrealtime) while the caller
 * runs the kernels under test on its own stream; _read waits for the monitor and returns the clock of every sampling
 * interval in MHz (delta cycles / delta reference x 100) -- the sustained clock under THAT load, which no host-side
 * sampler resolves for a 2 ms kernel.  The kernels under test are not instrumented. */
DAAM_API int daam_clock_monitor_start(DaamCtx* ctx, int n_samples, int period_us);
DAAM_API int daam_clock_monitor_read(DaamCtx* ctx, float* mhz, int capacity, int* n_intervals);

#ifdef __cplusplus
}
#endif
#endif /* DAAM_HIP_H */
