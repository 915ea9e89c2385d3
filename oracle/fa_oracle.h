/*
 * CPU oracle for the FlashAttention-2 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm (tile loop, masking rules,
 * online softmax, split-KV combine, KV-cache append + rotary, and the standard FA-2 backward
 * formulas).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product path (the HIP library) never links, imports or calls anything in this directory.
 *
 * Parity pin: checked against golden vectors produced in the build container from the
 * reference's own PyTorch oracle (attention_ref in /root/reference/tests/test_flash_attn.py)
 * and against torch SDPA; see tests/golden/make_golden.py and tests/test_oracle.py.
 *
 * All tensors are contiguous float32.  16-bit behaviour is modelled by rounding P, dS and the
 * outputs to fp16/bf16 (round-to-nearest-even) exactly where the reference converts.
 */
#ifndef FA_ORACLE_H
#define FA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { FA_ROUND_F32 = 0, FA_ROUND_F16 = 1, FA_ROUND_BF16 = 2 };

typedef struct fa_oracle_params {
    /* q: (b, seqlen_q, h, d), or (total_q, h, d) when cu_seqlens_q != NULL */
    const float *q;
    /* k, v: (b_cache, seqlen_k, h_k, d) | (total_k, h_k, d) when cu_seqlens_k is cumulative
     *       | (num_blocks, page_block_size, h_k, d) when block_table != NULL.
     * Non-const: the KV-cache append path writes knew/vnew into them.  */
    float *k;
    float *v;
    float *out;          /* same layout as q */
    float *lse;          /* (b, h, seqlen_q) */
    /* gradients (backward only), layouts of q/k/k/q */
    const float *dout;
    float *dq, *dk, *dv;
    float *softmax_d;    /* (b, h, seqlen_q) rowsum(dO * O), may be NULL */

    int b, h, h_k, d;
    int seqlen_q;        /* max query length (dense: the length) */
    int seqlen_k;        /* max key length / cache capacity */
    const int *cu_seqlens_q;   /* b+1 or NULL */
    const int *cu_seqlens_k;   /* b+1 cumulative, or b lengths when !is_seqlens_k_cumulative, or NULL */
    int is_seqlens_k_cumulative;
    const int *seqused_k;      /* b or NULL */
    const int *block_table;    /* (b, block_table_stride) or NULL */
    int block_table_stride;
    int page_block_size;
    const int *cache_batch_idx; /* b or NULL */
    const float *alibi_slopes;  /* (h) or (b, h) or NULL */
    int alibi_batch_stride;     /* 0 for (h) */
    float softmax_scale;
    int window_left, window_right; /* raw user values; folded by fa_oracle_fold_window */
    int is_causal;
    int round_mode;
    int block_m, block_n;       /* tile sizes; 0 = the reference's choice for d */
    int num_splits;             /* <=1: standard kernel; >1: split-KV kernel + combine */
    /* KV-cache append (fwd_kvcache with k/v): knew/vnew (b, seqlen_knew, h_k, d) */
    const float *knew, *vnew;
    int seqlen_knew;
    const float *rotary_cos, *rotary_sin; /* (seqlen_ro, rotary_dim/2) or NULL */
    int rotary_dim;
    int is_rotary_interleaved;
    /* Dropout (dropout.h:12-92, philox.cuh:12-49; compiled out of the fork, setup.py:204, but part
     * of the Python signatures).  p_dropout = probability of DROPPING (the user value); the Philox
     * seed / offset are what the generator handed out (flash_api.cpp:388-394). */
    float p_dropout;
    uint64_t philox_seed, philox_offset;
    /* Return_softmax output (flash_fwd_kernel.h:326-334): (b, h, seqlen_q_rounded, seqlen_k_rounded),
     * 16-bit-rounded exp(S - running max) with the sign bit set on dropped entries; NULL = off.
     * Only entries of visited tiles with row < seqlen_q and col < seqlen_k are written. */
    float *s_dmask;
    int seqlen_q_rounded, seqlen_k_rounded;
    /* rows of rotary_cos / rotary_sin (flash_api.cpp:876-877); > 0: the query's rotary position is
     * clamped to the last row (the reference kernel reads past the table when a caller appends
     * fewer keys than it has query rows under a causal mask; its own tests avoid that case,
     * tests/test_flash_attn.py:1955-1964) */
    int seqlen_ro;
} fa_oracle_params;

/* flash_api.cpp:289-294,121-124: window clamp, causal fold, one-sided fill. */
void fa_oracle_fold_window(int seqlen_q, int seqlen_k, int has_alibi, int *is_causal,
                           int *window_left, int *window_right, int *is_local);

/* flash_api.cpp:158-192 */
int fa_oracle_num_splits_heuristic(int batch_nheads_mblocks, int num_sms, int num_n_blocks,
                                   int max_splits);

/* flash_fwd_kernel.h:27-470 (standard) / :474-1053 + :1092-1250 (split + combine) */
int fa_oracle_fwd(fa_oracle_params *p);

/* Standard FA-2 backward (no kernel in the reference tree; flash.h:145-183 names the fields). */
int fa_oracle_bwd(fa_oracle_params *p);

float fa_oracle_round(float x, int round_mode);

/* philox.cuh:31-49 with the round count as a parameter (the reference runs 7): out = 4 words. */
void fa_oracle_philox(uint64_t seed, uint64_t subsequence, uint64_t offset, int n_rounds,
                      uint32_t out[4]);

/* dropout.h:17-23,39-79 + flash_fwd_kernel.h:324-325: 1 if attention entry (row, col) of
 * (batch bid, head hid) is KEPT.  keep_u8 = floor((1 - p_dropout) * 255) (flash_api.cpp:111). */
int fa_oracle_dropout_keep(uint64_t seed, uint64_t offset, int bid, int hid, int nheads, int row,
                           int col, int keep_u8);

/* Whole keep mask, (b, h, seqlen_q, seqlen_k) bytes. */
void fa_oracle_dropout_mask(uint64_t seed, uint64_t offset, int b, int h, int seqlen_q,
                            int seqlen_k, float p_dropout, uint8_t *mask);

#ifdef __cplusplus
}
#endif
#endif
