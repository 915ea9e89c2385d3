// Host side of the C ABI declared in include/vfa.h: argument checks, window/causal folding,
// kernel-parameter set-up and dispatch.  The counterpart of csrc/flash_attn/flash_api.cpp in the
// reference, minus everything that needs torch (that part lives in torch_binding.cpp).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vfa.h"
#include "vfa_kernel_params.h"
#include "vfa_dispatch.h"

namespace vfa {
hipError_t run_fwd_prefill(const KParams &p, hipStream_t stream);
hipError_t run_fwd_splitkv(const KParams &p, hipStream_t stream);
hipError_t run_append_kv(const KParams &p, hipStream_t stream);
hipError_t run_bwd(const KParams &p, hipStream_t stream, hipEvent_t *stage_events);
hipError_t run_fwd_prefill_dropout(const KParams &p, hipStream_t stream);
hipError_t run_fwd_prefill_split(const KParams &p, hipStream_t stream);
bool fwd_prefill_pack_ok(const KParams &p);                                  // fwd_prefill_pack.hip
hipError_t run_fwd_prefill_pack(const KParams &p, hipStream_t stream);
bool fwd_prefill_split_ok(const KParams &p);
hipError_t run_fwd_combine(const KParams &p, hipStream_t stream);
hipError_t run_bwd_dropout(const KParams &p, hipStream_t stream, hipEvent_t *stage_events);
}  // namespace vfa

using vfa::KParams;

#ifdef VFA_DEV
// Development build only (not part of include/vfa.h, absent from libvfa_hip.so): kernel-variant
// switches for A/B timing and for the tests that check a shipped kernel against its alternative.
namespace vfa { int g_tuning[8] = {0, 0, 0, 0, 0, 0, 0, 0}; }
extern "C" __attribute__((visibility("default"))) void vfa_debug_set_tuning(int idx, int value) {
    if (idx >= 0 && idx < 8) vfa::g_tuning[idx] = value;
}
#endif

static thread_local std::string g_err;

#define VFA_CHECK(cond, msg)      \
    do {                          \
        if (!(cond)) {            \
            g_err = (msg);        \
            return VFA_ERR_INVALID; \
        }                         \
    } while (0)

static int hip_fail(hipError_t e, const char *what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return VFA_ERR_HIP;
}

// The reference gates on sm8x/sm90 (flash_api.cpp:254-258); here the gate is gfx950.
static int check_arch() {
    static thread_local int cached_dev = -1;
    static thread_local int cached_ok = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    if (dev != cached_dev) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) return hip_fail(e, "hipGetDeviceProperties");
        cached_ok = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        cached_dev = dev;
    }
    if (!cached_ok) {
        g_err = "FlashAttention (MI355X build) only supports gfx950 GPUs.";
        return VFA_ERR_ARCH;
    }
    return VFA_OK;
}

static int num_cus() { return vfa::device_num_cus(); }

// Checks shared by the three forward entry points (flash_api.cpp:262-287 and siblings).
static int common_checks(const vfa_params *p) {
    VFA_CHECK(p != nullptr, "null params");
    VFA_CHECK(p->dtype == VFA_FP16 || p->dtype == VFA_BF16,
              "FlashAttention only support fp16 and bf16 data type");
    VFA_CHECK(p->q && p->k && p->v && p->out && p->softmax_lse, "null tensor pointer");
    VFA_CHECK(p->b > 0, "batch size must be postive");
    VFA_CHECK(p->d > 0 && p->d <= 256,
              "FlashAttention forward only supports head dimension at most 256");
    VFA_CHECK(p->d % 8 == 0, "head dimension must be a multiple of 8 (pad before calling)");
    VFA_CHECK(p->h_k > 0 && p->h % p->h_k == 0,
              "Number of heads in key/value must divide number of heads in query");
    VFA_CHECK(p->p_dropout >= 0.f && p->p_dropout < 1.f, "p_dropout must be in [0, 1)");  // flash_api.cpp:114
    VFA_CHECK(!p->s_dmask || p->p_dropout > 0.f, "return_softmax is only supported when p_dropout > 0.0");
    return VFA_OK;
}

// set_params_fprop (flash_api.cpp:21-138) + the folding rules in front of it (:289-294).
static void fill_common(const vfa_params *p, KParams &k, int seqlen_q_for_fold, int seqlen_k_for_fold) {
    std::memset(&k, 0, sizeof(k));
    k.is_bf16 = p->dtype == VFA_BF16;
    k.q = p->q; k.k = p->k; k.v = p->v; k.o = p->out; k.lse = p->softmax_lse;
    k.q_bs = p->q_batch_stride; k.q_rs = p->q_row_stride; k.q_hs = p->q_head_stride;
    k.k_bs = p->k_batch_stride; k.k_rs = p->k_row_stride; k.k_hs = p->k_head_stride;
    k.v_bs = p->v_batch_stride; k.v_rs = p->v_row_stride; k.v_hs = p->v_head_stride;
    k.o_bs = p->o_batch_stride; k.o_rs = p->o_row_stride; k.o_hs = p->o_head_stride;
    k.b = p->b; k.h = p->h; k.h_k = p->h_k; k.h_ratio = p->h / p->h_k; k.d = p->d;
    k.seqlen_q = p->seqlen_q; k.seqlen_k = p->seqlen_k;
    k.cu_seqlens_q = p->cu_seqlens_q; k.cu_seqlens_k = p->cu_seqlens_k;
    k.seqused_k = p->seqused_k;
    k.k_cumulative = p->cu_seqlens_k ? p->is_seqlens_k_cumulative : 1;
    k.block_table = p->block_table; k.bt_bs = p->block_table_batch_stride;
    k.page_size = p->block_table ? p->page_block_size : 1;
    k.page_shift = -1;
    if (k.page_size > 0 && (k.page_size & (k.page_size - 1)) == 0) {
        k.page_shift = 0;
        while ((1 << k.page_shift) < k.page_size) k.page_shift++;
    }
    k.cache_batch_idx = p->cache_batch_idx;
    k.alibi_slopes = p->alibi_slopes; k.alibi_bs = (int)p->alibi_slopes_batch_stride;
    k.scale = p->softmax_scale;
    k.scale_log2 = p->softmax_scale * (float)M_LOG2E;  // flash_api.cpp:103

    int wl = p->window_size_left, wr = p->window_size_right, causal = p->is_causal;
    if (wl >= seqlen_k_for_fold) wl = -1;              // flash_api.cpp:289-290
    if (wr >= seqlen_k_for_fold) wr = -1;
    if (seqlen_q_for_fold == 1 && !p->alibi_slopes) causal = 0;  // :293
    if (causal) wr = 0;                                // :294
    k.is_causal = (wl < 0 && wr == 0);                 // :121
    if (wl < 0 && wr >= 0) wl = seqlen_k_for_fold;     // :123-124
    if (wl >= 0 && wr < 0) wr = seqlen_k_for_fold;
    k.wl = wl; k.wr = wr;
    k.is_local = (wl >= 0 || wr >= 0) && !k.is_causal;
    k.dbg_flags = vfa::g_tuning[2];
    k.pack_gqa = 0;
    k.num_splits = 1;
    // flash_api.cpp:105-113: keep probability, its 8-bit threshold (rounded down: the test is <=)
    k.p_keep = 1.f - p->p_dropout;
    k.drop_keep_u8 = (int)(uint8_t)std::floor(k.p_keep * 255.0);
    k.rp_dropout = 1.f / k.p_keep;
    k.scale_rp = p->p_dropout > 0.f ? k.rp_dropout * k.scale : k.scale;
    k.philox_seed = p->philox_seed; k.philox_offset = p->philox_offset;
    k.seed_ptr = (const long long *)p->philox_seed_ptr; k.offset_ptr = (const long long *)p->philox_offset_ptr;
    k.offset_intragraph = p->philox_offset_intragraph;
    k.rng_state = (unsigned long long *)p->rng_state;
}

extern "C" int vfa_num_splits_heuristic(int batch_nheads_mblocks, int num_sms, int num_n_blocks,
                                        int max_splits) {
    // flash_api.cpp:158-192
    if (batch_nheads_mblocks >= 0.8f * num_sms) return 1;
    max_splits = std::min(std::min(max_splits, num_sms), num_n_blocks);
    float max_eff = 0.f;
    std::vector<float> eff;
    eff.reserve(max_splits > 0 ? max_splits : 1);
    auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
    auto eligible = [&](int s) { return s == 1 || cdiv(num_n_blocks, s) != cdiv(num_n_blocks, s - 1); };
    for (int s = 1; s <= max_splits; s++) {
        if (!eligible(s)) { eff.push_back(0.f); continue; }
        float n_waves = float(batch_nheads_mblocks * s) / num_sms;
        float e = n_waves / std::ceil(n_waves);
        if (e > max_eff) max_eff = e;
        eff.push_back(e);
    }
    for (int s = 1; s <= max_splits; s++) {
        if (!eligible(s)) continue;
        if (eff[s - 1] >= 0.85f * max_eff) return s;
    }
    return 1;
}

extern "C" int vfa_splitkv_num_splits(int batch, int num_heads, int head_size, int max_seqlen_k,
                                      int max_seqlen_q, int num_cus_) {
    // flash_api.cpp:199-210.  Up to 64 query rows (decode) the block sizes follow the reference,
    // so the split count is the same function of the shapes as the reference's.
    if (max_seqlen_q > 64) {
        // Beyond 64 query rows the split work items are the prefill kernel's 128-row blocks.  Their
        // fp32 partial results cost HBM traffic in proportion to rows x splits, so splitting is
        // kept for launches that would otherwise fill at most a quarter of the resident-workgroup
        // slots, and every split gets at least 16 tiles (1024 keys).  Measured on MI355X (h32/8
        // d128): 512 rows against 8192 keys, batch 1: 195 -> 116 us; 1024 x 1024: 30 -> 77 us (not
        // split under this rule).
        const int blocks = batch * num_heads * ((max_seqlen_q + 127) / 128);
        const int chunks = ((max_seqlen_k + 63) / 64) / 16;
        if (blocks * 4 > num_cus_ * 2 || chunks < 2) return 1;
        return vfa_num_splits_heuristic(blocks, num_cus_ * 2, chunks, 128);
    }
    const int block_n = head_size <= 64 ? 256 : (head_size <= 128 ? 128 : 64);
    const int num_n_blocks = (max_seqlen_k + block_n - 1) / block_n;
    const int num_m_blocks = (max_seqlen_q + 64 - 1) / 64;
    // The decode kernel cuts the keys in 128-key tiles whatever the head dim: a split count beyond
    // the number of tiles (possible above d = 128, where the reference's block is 64 keys) would
    // only add workgroups that write empty partials.
    const int tiles = std::max(1, (max_seqlen_k + 127) / 128);
    return std::min(tiles, vfa_num_splits_heuristic(batch * num_heads * num_m_blocks, num_cus_ * 2, num_n_blocks, 128));
}

// Grouped query heads with a few query positions per sequence -- more rows than one 32-row decode
// block holds, fewer positions than fill a 128-row prefill block per head: the prefill kernel with
// the group packed into the rows of its KV head (fwd_prefill_pack.hip) sweeps the keys once per KV
// head.  Measured (b64 kv4096 h32/8 d128 fp16, query positions 16 / 32 / 64): 260 / 456 / 597 us with
// 32-row decode blocks or one 128-row block per query head, 201 / 208 / 414 us packed.
static bool pack_rows_shape(int ratio, int seqlen_q) {
    if (vfa::g_tuning[2] & 4) return false;   // development switch 2, bit 2: never
    return ratio > 1 && seqlen_q * ratio > 32 && seqlen_q < 128;
}

extern "C" int vfa_decode_num_splits(int batch, int num_heads, int num_heads_k, int head_size, int max_seqlen_k,
                                     int max_seqlen_q, int num_cus_) {
    const int ratio = (num_heads_k > 0 && num_heads % num_heads_k == 0) ? num_heads / num_heads_k : 1;
    if (pack_rows_shape(ratio, max_seqlen_q)) {
        // work items: 128-row blocks of packed rows per (batch, KV head); the rule of the prefill
        // kernel's split variants (vfa_splitkv_num_splits above)
        const int blocks = batch * num_heads_k * ((max_seqlen_q * ratio + 127) / 128);
        const int chunks = ((max_seqlen_k + 63) / 64) / 4;   // at least 4 tiles (256 keys) per split
        if (blocks * 4 > num_cus_ * 2 || chunks < 2) return 1;
        return vfa_num_splits_heuristic(blocks, num_cus_ * 2, chunks, 128);
    }
    if (ratio <= 1 || max_seqlen_q > 64)
        return vfa_splitkv_num_splits(batch, num_heads, head_size, max_seqlen_k, max_seqlen_q, num_cus_);
    // run_fwd_splitkv / launch_decode_hpb: rows = seqlen_q * ratio in 32-row blocks, one (pair of)
    // KV head(s) per workgroup; a two-head workgroup fills a CU like two one-head workgroups, so the
    // count against 2 * CUs is per head either way
    const int m_blocks = (max_seqlen_q * ratio + 31) / 32;
    const int block_n = head_size <= 64 ? 256 : (head_size <= 128 ? 128 : 64);
    const int num_n_blocks = (max_seqlen_k + block_n - 1) / block_n;
    const int tiles = std::max(1, (max_seqlen_k + 127) / 128);
    return std::min(tiles, vfa_num_splits_heuristic(batch * num_heads_k * m_blocks, num_cus_ * 2, num_n_blocks, 128));
}

static int launch_fwd(const vfa_params *p, KParams &k, bool force_split) {
    hipStream_t stream = (hipStream_t)p->stream;
    hipError_t e;
    if (k.knew) {
        e = vfa::run_append_kv(k, stream);
        if (e != hipSuccess) return hip_fail(e, "append_kv launch");
        // the cache now holds the new rows; the attention kernels read it as a longer cache
    }
    if (p->p_dropout > 0.f) {
        // flash_api.cpp:206 (no split with dropout), :296 / :919 (no decode fold, no kv-cache path)
        VFA_CHECK(p->num_splits <= 1 && !force_split && !k.knew,
                  "dropout is not supported with split-KV, paged KV or KV-cache append");
        if (p->s_dmask) {
            VFA_CHECK(p->seqlen_q_rounded % 128 == 0 && p->seqlen_q_rounded >= k.seqlen_q &&
                      p->seqlen_k_rounded % 128 == 0 && p->seqlen_k_rounded >= k.seqlen_k,
                      "s_dmask needs seqlen_q_rounded / seqlen_k_rounded (multiples of 128)");
            k.s_dmask = p->s_dmask;
            k.sq_rounded_p = p->seqlen_q_rounded; k.sk_rounded_p = p->seqlen_k_rounded;
        }
        e = vfa::run_fwd_prefill_dropout(k, stream);
        if (e != hipSuccess) return hip_fail(e, "prefill (dropout) launch");
        return VFA_OK;
    }
    if (p->num_splits > 1) {
        VFA_CHECK(p->num_splits <= 128, "num_splits > 128 not supported");
        VFA_CHECK(p->oaccum && p->softmax_lseaccum, "split-KV needs oaccum / softmax_lseaccum scratch");
        k.num_splits = p->num_splits;
        k.oaccum = p->oaccum;
        k.lseaccum = p->softmax_lseaccum;
        // Few query rows: the decode kernel (waves split the keys of a 32-row block).  More: the
        // prefill kernel's split variants (128-row blocks), when they can load this K/V layout
        // (they do not rotate the query: rotary tables keep the decode kernel).
        const bool q_rotary = k.knew && k.rotary_dim > 0;
        if (pack_rows_shape(k.h_ratio, k.seqlen_q) && !q_rotary && vfa::fwd_prefill_pack_ok(k)) {
            e = vfa::run_fwd_prefill_pack(k, stream);
            if (e == hipSuccess) e = vfa::run_fwd_combine(k, stream);
        } else if (k.seqlen_q > 64 && !q_rotary && vfa::fwd_prefill_split_ok(k)) {
            e = vfa::run_fwd_prefill_split(k, stream);
            if (e == hipSuccess) e = vfa::run_fwd_combine(k, stream);
        } else {
            e = vfa::run_fwd_splitkv(k, stream);
        }
        if (e != hipSuccess) return hip_fail(e, "splitkv launch");
        return VFA_OK;
    }
    // A few query positions of grouped heads: the group packed into the rows of its KV head (the
    // prefill kernel does not rotate the query: rotary tables keep the decode kernel).
    // ... and ragged batches whose sequences are short on average (decode sequences and short chunks
    // batched with a few long ones: a sequence of one query position then costs one workgroup per KV
    // head instead of one per query head) or that attend through a sliding window (a packed block
    // spans a quarter of the positions, so fewer of its tiles lie on the diagonal and on the
    // window's edge).  Measured on MI355X, h32/8 d128, one ragged call (tools/perf_mixed.py and a
    // batch of 8192 ... 127 tokens): 60 x 1 + 4 x 512 positions 722 -> 347 us; 80 x 300 causal 198 ->
    // 181; window 1024 + ALiBi 610 -> 537, without ALiBi 497 -> 478; window 4096 1022 -> 1002; 24 x 1024
    // causal level; long causal sequences 1.6 % behind (they keep one block per query head).
    // (Above d = 128 the unpacked rows have the hand-pipelined kernel, the packed ones the general
    // one: there only batches of mostly one-block sequences are packed.)
    const long long avg_lim = k.d <= 128 ? 512 : 128;
    const bool short_ragged = k.h_ratio > 1 && !(vfa::g_tuning[2] & 4) && p->cu_seqlens_q && k.seqlen_q >= 128 &&
                              ((p->total_q > 0 && (long long)p->total_q < avg_lim * k.b) || (k.is_local && k.d <= 128) ||
                               (vfa::g_tuning[2] & 8));
    if ((pack_rows_shape(k.h_ratio, k.seqlen_q) || short_ragged) && !(k.knew && k.rotary_dim > 0) &&
        vfa::fwd_prefill_pack_ok(k)) {
        e = vfa::run_fwd_prefill_pack(k, stream);
        if (e != hipSuccess) return hip_fail(e, "prefill (packed heads) launch");
        return VFA_OK;
    }
    // Small query blocks against long keys (decode): the split kernel's tiling (few query rows,
    // many waves along the keys) is the right one even with a single split.  The reference sends
    // every call that appends, pages or indexes the cache to its split kernel (flash_api.cpp:919)
    // because only that kernel can; here the prefill kernels read paged and indexed caches too and
    // the new keys are in the cache before the attention kernel starts, so only a query that is
    // still to be rotated (the decode kernel does that on the fly) keeps long query blocks there
    // (measured b64 kv4096 h32/8 d128 paged, 64 query positions: 1049 us in 32-row decode blocks).
    const bool q_rotary = k.knew && k.rotary_dim > 0;
    (void)force_split;
    if (q_rotary || k.seqlen_q <= 32) {
        e = vfa::run_fwd_splitkv(k, stream);
        if (e != hipSuccess) return hip_fail(e, "decode launch");
        return VFA_OK;
    }
    e = vfa::run_fwd_prefill(k, stream);
    if (e != hipSuccess) return hip_fail(e, "prefill launch");
    return VFA_OK;
}

extern "C" int vfa_mha_fwd(const vfa_params *p) {
    g_err.clear();
    int rc = common_checks(p);
    if (rc) return rc;
    if ((rc = check_arch())) return rc;
    VFA_CHECK(!p->cu_seqlens_q && !p->cu_seqlens_k, "vfa_mha_fwd takes dense batches");
    VFA_CHECK(p->seqlen_q > 0, "seqlen_q must be positive");
    KParams k;
    fill_common(p, k, p->seqlen_q, p->seqlen_k);
    return launch_fwd(p, k, false);
}

extern "C" int vfa_mha_varlen_fwd(const vfa_params *p) {
    g_err.clear();
    int rc = common_checks(p);
    if (rc) return rc;
    if ((rc = check_arch())) return rc;
    // cu_seqlens_q may be NULL for a dense query batch (single-token GQA decode folded into rows,
    // flash_api.cpp:500-507)
    VFA_CHECK(p->cu_seqlens_k, "cu_seqlens_k is required");
    VFA_CHECK(!p->block_table || p->page_block_size % 16 == 0,
              "Paged KV cache block size must be divisible by 16");
    KParams k;
    // flash_api.cpp:493-494,515-516: the varlen entry folds with the MAX lengths
    fill_common(p, k, p->seqlen_q, p->seqlen_k);
    k.k_cumulative = 1;
    return launch_fwd(p, k, p->p_dropout > 0.f && p->block_table);
}

extern "C" int vfa_mha_fwd_kvcache(const vfa_params *p) {
    g_err.clear();
    int rc = common_checks(p);
    if (rc) return rc;
    if ((rc = check_arch())) return rc;
    VFA_CHECK(!p->cu_seqlens_q, "fwd_kvcache takes a dense query batch");
    VFA_CHECK(p->p_dropout == 0.f, "fwd_kvcache has no dropout");
    VFA_CHECK(!(p->block_table && p->cache_batch_idx), "Paged KVcache does not support cache_batch_idx");
    VFA_CHECK(!p->block_table || p->page_block_size % 16 == 0,
              "Paged KV cache block size must be divisible by 16");
    KParams k;
    fill_common(p, k, p->seqlen_q, p->seqlen_k);
    k.k_cumulative = p->cu_seqlens_k ? 0 : 1;  // flash_api.cpp:859-867: lengths, not offsets
    if (p->knew) {
        VFA_CHECK(p->vnew, "If key is supplied, value must also be passed in");
        VFA_CHECK(p->cu_seqlens_k, "If key is supplied, seqlens_k must also be passed in");
        VFA_CHECK(p->seqlen_q <= p->seqlen_k,
                  "If key is supplied, it must have seqlen <= the seqlen of the KV cache");
        k.knew = p->knew; k.vnew = p->vnew; k.seqlen_knew = p->seqlen_knew;
        k.knew_bs = p->knew_batch_stride; k.knew_rs = p->knew_row_stride; k.knew_hs = p->knew_head_stride;
        k.vnew_bs = p->vnew_batch_stride; k.vnew_rs = p->vnew_row_stride; k.vnew_hs = p->vnew_head_stride;
    }
    if (p->rotary_cos) {
        VFA_CHECK(p->knew, "If rotary cos/sin are provided, new key / value to be appended to KV cache must also be provided");
        VFA_CHECK(p->rotary_sin, "If rotary cos is provided, rotary sin must also be provided");
        VFA_CHECK(p->rotary_dim <= p->d, "rotary_dim must be <= headdim");
        VFA_CHECK(p->rotary_dim % 16 == 0, "Only rotary dimensions divisible by 16 are currently supported");
        k.rotary_cos = p->rotary_cos; k.rotary_sin = p->rotary_sin;
        k.rotary_dim = p->rotary_dim; k.rotary_interleaved = p->is_rotary_interleaved;
        VFA_CHECK(p->seqlen_ro == 0 || p->seqlen_ro >= p->seqlen_k, "cos/sin seqlen must be at least the seqlen of KV cache");
        k.seqlen_ro = p->seqlen_ro;
    }
    // flash_api.cpp:919: appending, cache_batch_idx and paged caches take the split kernel
    const bool force = p->knew || p->cache_batch_idx || p->block_table;
    return launch_fwd(p, k, force);
}

// ---------------------------------------------------------------------------------------------
// vfa_mha_fwd_kvcache_noalloc: the split-KV partial results live in a file-scope device array
// instead of caller-provided workspace, for callers that cannot allocate (HIP graph capture, tight
// serving loops).  Every GPU the library runs on holds its own copy of the array at its own
// address until the process ends; in host code the array's NAME is only a handle for
// hipGetSymbolAddress -- the address kernels and copies use is the per-device pointer below.
static __device__ __align__(256) float scratch[VFA_NOALLOC_SCRATCH_BYTES / sizeof(float)];

static constexpr int kMaxDevices = 64;
static std::atomic<float *> g_scratch_ptr[kMaxDevices];

// Device address of `scratch` on the current device, resolved once per device (with that device
// current, as it is for the caller's launch).
static int scratch_on_current_device(float **out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    if (dev < 0 || dev >= kMaxDevices) {
        g_err = "vfa_mha_fwd_kvcache_noalloc: device index beyond the per-device scratch table";
        return VFA_ERR_UNSUPPORTED;
    }
    float *ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!ptr) {
        e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return hip_fail(e, "hipGetSymbolAddress(scratch)");
        if (!ptr) {
            g_err = "hipGetSymbolAddress(scratch) returned a null address";
            return VFA_ERR_HIP;
        }
        g_scratch_ptr[dev].store(ptr, std::memory_order_release);
    }
    *out = ptr;
    return VFA_OK;
}

static size_t split_scratch_floats(const vfa_params *p, int splits, size_t *lse_floats) {
    const size_t rows = (size_t)p->b * (size_t)p->h * (size_t)p->seqlen_q;
    const size_t lse = ((size_t)splits * rows + 63) / 64 * 64;  // keeps the O partials 256-byte aligned
    if (lse_floats) *lse_floats = lse;
    return lse + (size_t)splits * rows * (size_t)p->d;
}

extern "C" int vfa_mha_fwd_kvcache_noalloc(const vfa_params *p) {
    g_err.clear();
    int rc = common_checks(p);
    if (rc) return rc;
    VFA_CHECK(p->seqlen_q > 0, "seqlen_q must be positive");
    vfa_params q = *p;
    q.oaccum = nullptr;
    q.softmax_lseaccum = nullptr;
    constexpr size_t cap = VFA_NOALLOC_SCRATCH_BYTES / sizeof(float);
    if (q.num_splits == 0) {
        // the count the allocating path's caller would pick (set_params_splitkv), lowered until the
        // partial results fit the static array
        q.num_splits = vfa_decode_num_splits(p->b, p->h, p->h_k, p->d, p->seqlen_k, p->seqlen_q, num_cus());
        while (q.num_splits > 1 && split_scratch_floats(p, q.num_splits, nullptr) > cap) q.num_splits--;
    }
    if (q.num_splits > 1) {
        VFA_CHECK(q.num_splits <= 128, "num_splits > 128 not supported");
        size_t lse_floats = 0;
        const size_t need = split_scratch_floats(p, q.num_splits, &lse_floats);
        if (need > cap) {
            g_err = "vfa_mha_fwd_kvcache_noalloc: num_splits * b * h * seqlen_q * (d + 1) floats = " +
                    std::to_string(need * sizeof(float)) + " bytes exceed the static scratch of " +
                    std::to_string((size_t)VFA_NOALLOC_SCRATCH_BYTES) +
                    " bytes; lower num_splits or call vfa_mha_fwd_kvcache with caller-provided scratch";
            return VFA_ERR_INVALID;
        }
        float *ptr = nullptr;
        if ((rc = scratch_on_current_device(&ptr))) return rc;
        // Every split of every row is written by the kernels before the combine kernel reads it
        // (empty splits store 0 / -inf), so nothing has to start at zero; the previous call's
        // contents are simply overwritten.  Two calls that overlap on one GPU would race here.
        q.softmax_lseaccum = ptr;
        q.oaccum = ptr + lse_floats;
    }
    return vfa_mha_fwd_kvcache(&q);
}

extern "C" size_t vfa_bwd_dq_accum_elems(int b, int seqlen_q, int h, int d, int total_q_or_0) {
    const size_t rows = total_q_or_0 > 0 ? (size_t)total_q_or_0 : (size_t)b * seqlen_q;
    return rows * (size_t)h * (size_t)d;
}

extern "C" int vfa_bwd_head_splits(int batch, int num_heads, int num_heads_k, int max_seqlen_k, int num_cus_) {
    // The dK/dV kernel's work items are (batch, kv head, 256-key block); aim at two per CU.
    if (num_heads_k <= 0 || num_heads % num_heads_k != 0) return 1;
    const int g = num_heads / num_heads_k;
    const long long blocks = (long long)((max_seqlen_k + 255) / 256) * batch * num_heads_k;
    const long long target = 2LL * num_cus_;
    if (g == 1 || blocks <= 0 || blocks >= target) return 1;
    return (int)std::min<long long>(g, (target + blocks - 1) / blocks);
}

static int bwd_common(const vfa_params *p, bool varlen) {
    g_err.clear();
    int rc = common_checks(p);
    if (rc) return rc;
    if ((rc = check_arch())) return rc;
    VFA_CHECK(p->dout && p->dq && p->dk && p->dv, "null gradient pointer");
    VFA_CHECK(p->dsoftmax_sum, "backward needs the dsoftmax_sum buffer");
    VFA_CHECK(varlen == (p->cu_seqlens_q != nullptr), "cu_seqlens mismatch for this entry point");
    KParams k;
    fill_common(p, k, p->seqlen_q, p->seqlen_k);
    k.k_cumulative = 1;
    k.dout = p->dout; k.dq = p->dq; k.dk = p->dk; k.dv = p->dv;
    k.do_bs = p->do_batch_stride; k.do_rs = p->do_row_stride; k.do_hs = p->do_head_stride;
    k.dq_bs = p->dq_batch_stride; k.dq_rs = p->dq_row_stride; k.dq_hs = p->dq_head_stride;
    k.dk_bs = p->dk_batch_stride; k.dk_rs = p->dk_row_stride; k.dk_hs = p->dk_head_stride;
    k.dv_bs = p->dv_batch_stride; k.dv_rs = p->dv_row_stride; k.dv_hs = p->dv_head_stride;
    k.dsoftmax_sum = p->dsoftmax_sum;
    k.dq_accum = p->dq_accum;
    k.deterministic = p->deterministic;
    k.seqlen_q_rounded = (p->seqlen_q + 127) / 128 * 128;
    k.rng_read = p->rng_state != nullptr;
    if (p->bwd_head_splits > 1) {
        VFA_CHECK(p->dk_accum && p->dv_accum && p->bwd_part_rows > 0,
                  "bwd_head_splits > 1 needs dk_accum / dv_accum slabs and bwd_part_rows");
        VFA_CHECK(p->bwd_head_splits <= p->h / p->h_k, "bwd_head_splits exceeds the heads per KV head");
        k.hsplit = p->bwd_head_splits;
        k.dk_part = p->dk_accum; k.dv_part = p->dv_accum;
        k.part_rows = p->bwd_part_rows;
        k.part_stride = p->bwd_part_rows * p->h_k * p->d;
    }
    hipError_t e = p->p_dropout > 0.f
                       ? vfa::run_bwd_dropout(k, (hipStream_t)p->stream, (hipEvent_t *)p->stage_events)
                       : vfa::run_bwd(k, (hipStream_t)p->stream, (hipEvent_t *)p->stage_events);
    if (e != hipSuccess) return hip_fail(e, "backward launch");
    return VFA_OK;
}

extern "C" int vfa_mha_bwd(const vfa_params *p) { return bwd_common(p, false); }
extern "C" int vfa_mha_varlen_bwd(const vfa_params *p) { return bwd_common(p, true); }

extern "C" int vfa_dispatch_geometry(const vfa_params *p, int kernel) {
    g_err.clear();
    int rc = common_checks(p);
    if (rc) return rc;
    VFA_CHECK(kernel >= 0 && kernel <= 2, "kernel: 0 = forward, 1 = dK/dV, 2 = dQ");
    KParams k;
    fill_common(p, k, p->seqlen_q, p->seqlen_k);
    k.k_cumulative = 1;
    k.dout = p->dout;
    k.do_rs = p->do_row_stride;
    k.hsplit = p->bwd_head_splits > 1 ? p->bwd_head_splits : 0;
    const int cus = num_cus();
    return kernel == 0 ? vfa::fwd_geometry(k, cus) : kernel == 1 ? vfa::dkdv_geometry(k, cus) : vfa::dq_geometry(k, cus);
}

extern "C" size_t vfa_params_sizeof(void) { return sizeof(vfa_params); }
extern "C" const char *vfa_last_error(void) { return g_err.c_str(); }
extern "C" const char *vfa_version(void) { return "vfa-mi355x 0.1.0 (api of vllm_flash_attn 2.5.8.post3)"; }
