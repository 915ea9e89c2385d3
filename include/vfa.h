/*
 * vfa.h -- C ABI of the MI355X (gfx950) FlashAttention-2 library, libvfa_hip.so.
 *
 * This is the drop-in boundary of the hot path: plain pointers, sizes and strides, no torch types.
 * The PyTorch extension module `vllm_flash_attn_2_cuda` (fwd / varlen_fwd / fwd_kvcache / bwd /
 * varlen_bwd) is a thin shim that validates tensors, allocates outputs and calls these entry
 * points; any other host (ctypes, cgo, JNI, ...) can bind them directly.
 *
 * Conventions
 *   - all data pointers are DEVICE pointers on the current HIP device; `stream` is a hipStream_t
 *     (NULL = the default stream).  Calls enqueue work and return; they never synchronise and
 *     never allocate device memory.
 *   - strides are in ELEMENTS; the last (head_dim) stride must be 1.
 *   - dtype: VFA_FP16 or VFA_BF16 for q/k/v/out/dout/dq/dk/dv; LSE and accumulators are fp32.
 *   - return value: 0 on success, otherwise a negative vfa_status; vfa_last_error() returns the
 *     message (the reference's TORCH_CHECK strings where one exists) for the calling thread.
 */
#ifndef VFA_H
#define VFA_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with hidden visibility: the entry points declared here are its whole
 * dynamic symbol table. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define VFA_FP16 0
#define VFA_BF16 1

typedef enum vfa_status {
    VFA_OK = 0,
    VFA_ERR_INVALID = -1,      /* failed argument check (message = the reference's check text) */
    VFA_ERR_UNSUPPORTED = -2,  /* valid for the reference, not built here */
    VFA_ERR_HIP = -3,          /* HIP runtime / launch error */
    VFA_ERR_ARCH = -4          /* device is not gfx950 */
} vfa_status;

/* One argument block for every entry point (unused fields zero / NULL).
 * Field meaning follows the reference's Flash_fwd_params / Flash_bwd_params
 * (csrc/flash_attn/src/flash.h:24-183) as filled by set_params_fprop
 * (csrc/flash_attn/flash_api.cpp:21-138). */
typedef struct vfa_params {
    int dtype;                       /* VFA_FP16 | VFA_BF16 */
    /* q: (b, seqlen_q, h, d) or (total_q, h, d) when cu_seqlens_q is set */
    const void *q, *k, *v;
    void *out;                       /* same shape as q */
    float *softmax_lse;              /* (b, h, seqlen_q) fp32; varlen: (b, h, max_seqlen_q) */
    int64_t q_batch_stride, q_row_stride, q_head_stride;
    int64_t k_batch_stride, k_row_stride, k_head_stride;   /* paged: batch stride = page stride */
    int64_t v_batch_stride, v_row_stride, v_head_stride;
    int64_t o_batch_stride, o_row_stride, o_head_stride;
    int b, seqlen_q, seqlen_k;       /* seqlen_*: max lengths (cache capacity for kv-cache) */
    int h, h_k, d;
    const int32_t *cu_seqlens_q;     /* (b+1) or NULL */
    const int32_t *cu_seqlens_k;     /* (b+1), or (b) lengths when !is_seqlens_k_cumulative */
    const int32_t *seqused_k;        /* (b) or NULL */
    int is_seqlens_k_cumulative;
    const int32_t *block_table;      /* (b, max_blocks) or NULL */
    int64_t block_table_batch_stride;
    int page_block_size;
    const int32_t *cache_batch_idx;  /* (b) or NULL */
    const float *alibi_slopes;       /* (h) or (b, h) fp32, or NULL */
    int64_t alibi_slopes_batch_stride; /* 0 for (h) */
    float p_dropout;                 /* probability of DROPPING an attention entry, in [0, 1); > 0 only
                                        for vfa_mha_fwd / vfa_mha_varlen_fwd (no split, no paging) and
                                        the backward entry points (flash_api.cpp:105-117) */
    float softmax_scale;
    int is_causal;
    int window_size_left, window_size_right;   /* raw user values; folded as flash_api.cpp:289-294 */
    /* split-KV (fwd_kvcache / decode): caller-provided fp32 scratch when num_splits > 1 */
    int num_splits;                  /* 0/1 = no split */
    float *softmax_lseaccum;         /* (num_splits, b, h, seqlen_q) */
    float *oaccum;                   /* (num_splits, b, h, seqlen_q, d) */
    /* KV-cache append: k/v new (b, seqlen_knew, h_k, d), rotary cos/sin (seqlen_ro, rotary_dim/2) */
    const void *knew, *vnew;
    int64_t knew_batch_stride, knew_row_stride, knew_head_stride;
    int64_t vnew_batch_stride, vnew_row_stride, vnew_head_stride;
    int seqlen_knew;
    const void *rotary_cos, *rotary_sin;
    int rotary_dim;
    int is_rotary_interleaved;
    /* backward */
    const void *dout;
    void *dq, *dk, *dv;              /* dk/dv: (b, seqlen_k, h_k, d) (summed over the GQA group) */
    int64_t do_batch_stride, do_row_stride, do_head_stride;
    int64_t dq_batch_stride, dq_row_stride, dq_head_stride;
    int64_t dk_batch_stride, dk_row_stride, dk_head_stride;
    int64_t dv_batch_stride, dv_row_stride, dv_head_stride;
    float *dsoftmax_sum;             /* (b, h, seqlen_q rounded up to 128) fp32: rowsum(dO * O) */
    float *dq_accum;                 /* unused, may be NULL: kept for the field list of flash.h:153-156
                                        (dQ is accumulated in registers by a query-stationary kernel) */
    float *dk_accum, *dv_accum;      /* fp32 partial dK / dV slabs when bwd_head_splits > 1 (below); otherwise
                                        unused, may be NULL (dK/dV of a GQA group are summed in registers) */
    int deterministic;
    void *stream;                    /* hipStream_t */
    /* Optional instrumentation (NULL = off): array of 4 hipEvent_t that vfa_mha_bwd /
     * vfa_mha_varlen_bwd record on `stream` before the preprocess kernel, before the main kernel,
     * after the main kernel and after the dq conversion kernel, so a caller can time the main
     * kernel alone.  No counterpart in the reference. */
    void **stage_events;
    /* Dropout (csrc/flash_attn/src/dropout.h:12-92, philox.cuh:12-49; flash_api.cpp:379-394).
     * Forward: the Philox seed / offset the caller's generator handed out (the offset must then
     * advance by b * h * 32); if rng_state != NULL the kernel also writes {seed, offset} there
     * (2 x uint64 on the device, flash_fwd_kernel.h:51-54).  Backward: rng_state != NULL is READ on
     * the device for {seed, offset} (what the forward wrote), else philox_seed / philox_offset. */
    uint64_t philox_seed, philox_offset;
    uint64_t *rng_state;
    /* Forward under graph capture: when philox_seed_ptr != NULL the kernel reads the seed from
     * *philox_seed_ptr and the offset from *philox_offset_ptr + philox_offset_intragraph ON THE
     * DEVICE, at replay time, instead of philox_seed / philox_offset -- the captured form of a
     * generator's state (at::PhiloxCudaState; the reference passes it as params.philox_args,
     * flash_api.cpp:388-394, and unpacks it in the kernel, flash_fwd_kernel.h:45). */
    const int64_t *philox_seed_ptr, *philox_offset_ptr;
    uint32_t philox_offset_intragraph;
    /* Return_softmax output (flash_fwd_kernel.h:326-334), forward with p_dropout > 0 only: 16-bit
     * (b, h, seqlen_q_rounded, seqlen_k_rounded), exp(score - running max) per 64-key tile with
     * the sign bit set on dropped entries; NULL = off.  Both rounded lengths are multiples of 128
     * (flash_api.cpp:339-340,353).  For testing, like the reference's. */
    void *s_dmask;
    int seqlen_q_rounded, seqlen_k_rounded;
    /* Backward head split (no counterpart in the reference, which has no backward kernel): with
     * bwd_head_splits > 1 the query heads of a KV group are swept by that many workgroups per key
     * block, which leave fp32 partial dK / dV in dk_accum / dv_accum -- each
     * (bwd_head_splits, bwd_part_rows, h_k, d) contiguous, bwd_part_rows = b * seqlen_k for dense
     * batches, total_k for ragged ones -- and a reduce kernel adds the slabs in index order (the
     * result stays bitwise reproducible).  For launches with too few (batch, kv head, key block)
     * items to fill the chip; vfa_bwd_head_splits() suggests the count.  0 / 1 = off. */
    int bwd_head_splits;
    int64_t bwd_part_rows;
    /* Rows of rotary_cos / rotary_sin (flash_api.cpp:876-877: must be >= seqlen_k).  A query row is
     * rotated at position cache_seqlen (+ its row index under a causal or window mask), which a
     * caller can push past seqlen_k by appending fewer new keys than it has query rows; with
     * seqlen_ro > 0 that position is clamped to the table's last row instead of being read past
     * its end.  0 = unknown: no clamp and no check (the reference kernel's behaviour). */
    int seqlen_ro;
    /* Ragged forward: the rows of q (the reference's total_q, flash_api.cpp:475).  A hint for the
     * dispatch -- with grouped heads and fewer than 512 rows per sequence on average (decode
     * sequences and short chunks batched with a few long ones, as a serving engine batches them) the
     * query heads of a KV head are packed into the rows of one workgroup.  0 = unknown. */
    int total_q;
} vfa_params;

/* Forward, dense batch.  Replaces mha_fwd, csrc/flash_attn/flash_api.cpp:240-420
 * (pybind name `fwd`, :941). */
int vfa_mha_fwd(const vfa_params *p);

/* Forward, ragged batch (cu_seqlens) with optional paged K/V.  Replaces mha_varlen_fwd,
 * csrc/flash_attn/flash_api.cpp:422-669 (pybind name `varlen_fwd`, :942). */
int vfa_mha_varlen_fwd(const vfa_params *p);

/* Forward against a KV cache (decode): cache_seqlens, cache_batch_idx, paged cache, optional
 * in-place append of knew/vnew with rotary, optional split-KV.  Replaces mha_fwd_kvcache,
 * csrc/flash_attn/flash_api.cpp:671-937 (pybind name `fwd_kvcache`, :945). */
int vfa_mha_fwd_kvcache(const vfa_params *p);

/* vfa_mha_fwd_kvcache for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference, whose
 * split scratch is two torch::empty calls inside set_params_splitkv (flash_api.cpp:212-215).
 * Same arguments and results as vfa_mha_fwd_kvcache, except that softmax_lseaccum / oaccum are
 * ignored: the split-KV partial results go to a static device array of
 * VFA_NOALLOC_SCRATCH_BYTES that the library holds on every GPU it runs on.  num_splits == 0 picks
 * the count like vfa_splitkv_num_splits, lowered until it fits; an explicit count whose partials
 * (num_splits * b * h * seqlen_q * (d + 1) floats) exceed the array is rejected with
 * VFA_ERR_INVALID.  Calls no hipMalloc / hipFree or other allocating routine.  All calls on one GPU
 * share that GPU's array: calls that overlap in time on one GPU (different streams) race. */
#define VFA_NOALLOC_SCRATCH_BYTES (32u << 20)
int vfa_mha_fwd_kvcache_noalloc(const vfa_params *p);

/* Backward, dense batch.  The reference fork exports no backward (flash_api.cpp:943); the
 * argument list is pinned by its Python call site vllm_flash_attn/flash_attn_interface.py:135-154
 * and the parameter block csrc/flash_attn/src/flash.h:145-183. */
int vfa_mha_bwd(const vfa_params *p);

/* Backward, ragged batch.  Call site vllm_flash_attn/flash_attn_interface.py:183-207. */
int vfa_mha_varlen_bwd(const vfa_params *p);

/* Split count for the split-KV path.  Replaces num_splits_heuristic,
 * csrc/flash_attn/flash_api.cpp:158-192. */
int vfa_num_splits_heuristic(int batch_nheads_mblocks, int num_sms, int num_n_blocks, int max_splits);

/* Split count a forward call with these dims would pick when num_splits == 0, given the device's
 * CU count: the sizing half of set_params_splitkv, csrc/flash_attn/flash_api.cpp:194-219. */
int vfa_splitkv_num_splits(int batch, int num_heads, int head_size, int max_seqlen_k,
                           int max_seqlen_q, int num_cus);

/* The same for grouped query heads (num_heads > num_heads_k): up to 64 query rows the decode
 * kernel packs the query heads of a KV head into the rows of one workgroup, so the work items the
 * split count has to fill the GPU with are (batch, KV head, 32-row block of seqlen_q * group rows)
 * -- this function counts them the way the kernel is launched; with num_heads == num_heads_k or
 * more than 64 query rows it equals vfa_splitkv_num_splits.  What a caller that passes
 * num_splits == 0 semantics (flash_api.cpp:207-210) should size its scratch with. */
int vfa_decode_num_splits(int batch, int num_heads, int num_heads_k, int head_size, int max_seqlen_k,
                          int max_seqlen_q, int num_cus);

/* Elements of the fp32 dq_accum buffer of the reference's parameter block
 * (csrc/flash_attn/src/flash.h:153-156); this library does not need the buffer. */
size_t vfa_bwd_dq_accum_elems(int b, int seqlen_q, int h, int d, int total_q_or_0);

/* Head-split count the backward should use for these dims on a device with num_cus compute units
 * (1 = none): see vfa_params.bwd_head_splits. */
int vfa_bwd_head_splits(int batch, int num_heads, int num_heads_k, int max_seqlen_k, int num_cus);

/* Which kernel geometry a call with these arguments takes (a diagnostic: nothing is launched and
 * no GPU is needed; the device's CU count is 256 when none is current).  `kernel`: 0 = forward
 * (vfa_mha_fwd / vfa_mha_varlen_fwd prefill path), 1 = backward dK/dV kernel, 2 = backward dQ
 * kernel.  Returns a VFA_GEO_* code or a negative vfa_status.  The counterpart of the reference's
 * tile table and CTA-count test (csrc/flash_attn/src/flash_fwd_launch_template.h:168-370,
 * csrc/flash_attn/flash_api.cpp:158-219), which the reference does not expose. */
#define VFA_GEO_BLOCK4 4     /* one 4-wave workgroup per 128-row (128-key) block, two per CU */
#define VFA_GEO_BLOCK8 8     /* one 8-wave workgroup per 256-row (256-key) block */
#define VFA_GEO_PERSIST8 18  /* one 8-wave workgroup per CU walks 256-row blocks */
#define VFA_GEO_WAVE64 64    /* dK/dV: 4 waves x 64 keys, one wave per SIMD */
#define VFA_GEO_WIDE32 32    /* d > 128: 4 waves at one wave per SIMD, accumulators at fixed registers (forward: 32 rows per
                                wave, hand-pipelined; dK/dV: 32 keys x all 256 columns per wave) */
int vfa_dispatch_geometry(const vfa_params *p, int kernel);

/* sizeof(vfa_params) of the built library, for foreign-function bindings to check their mirror of
 * the struct against (no counterpart in the reference: its boundary is typed by pybind). */
size_t vfa_params_sizeof(void);

/* Message of the last failed call on this thread ("" if none). */
const char *vfa_last_error(void);

/* Library version string. */
const char *vfa_version(void);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* VFA_H */
