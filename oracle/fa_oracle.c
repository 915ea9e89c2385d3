/*
 * CPU oracle for the FlashAttention-2 hot path.  TEST INFRASTRUCTURE ONLY (see fa_oracle.h).
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 */
#include "fa_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXD 256
static const float LOG2E = 1.4426950408889634f;

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

/* ---- 16-bit rounding (round-to-nearest-even), as cutlass NumericConverter does for
 * csrc/flash_attn/src/utils.h:227-235 (convert_type). ---- */
static float round_bf16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return x; /* NaN stays NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&x, &u, 4);
    return x;
}

static float round_f16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7fffffffu;
    float r;
    if (a > 0x7f800000u) return x;                 /* NaN */
    if (a >= 0x477ff000u) {                        /* >= 65520 rounds to inf */
        uint32_t inf = sign | 0x7f800000u;
        memcpy(&r, &inf, 4);
        return r;
    }
    if (a < 0x38800000u) {                         /* subnormal half: quantum 2^-24 */
        float ax;
        memcpy(&ax, &a, 4);
        float q = rintf(ax * 16777216.0f) / 16777216.0f; /* rintf = RNE in default mode */
        uint32_t qu;
        memcpy(&qu, &q, 4);
        qu |= sign;
        memcpy(&r, &qu, 4);
        return r;
    }
    a += 0xfffu + ((a >> 13) & 1u);
    a &= 0xffffe000u;
    a |= sign;
    memcpy(&r, &a, 4);
    return r;
}

float fa_oracle_round(float x, int mode) {
    if (mode == FA_ROUND_F16) return round_f16(x);
    if (mode == FA_ROUND_BF16) return round_bf16(x);
    return x;
}

/* csrc/flash_attn/flash_api.cpp:289-294 (clamp, sq==1 causal drop, causal fold) and
 * :121-124 (is_causal, one-sided window fill) */
void fa_oracle_fold_window(int seqlen_q, int seqlen_k, int has_alibi, int *is_causal, int *wl,
                           int *wr, int *is_local) {
    if (*wl >= seqlen_k) *wl = -1;
    if (*wr >= seqlen_k) *wr = -1;
    if (seqlen_q == 1 && !has_alibi) *is_causal = 0;
    if (*is_causal) *wr = 0;
    *is_causal = (*wl < 0 && *wr == 0);
    if (*wl < 0 && *wr >= 0) *wl = seqlen_k;
    if (*wl >= 0 && *wr < 0) *wr = seqlen_k;
    *is_local = (*wl >= 0 || *wr >= 0) && !*is_causal;
}

/* csrc/flash_attn/flash_api.cpp:158-192 */
int fa_oracle_num_splits_heuristic(int batch_nheads_mblocks, int num_sms, int num_n_blocks,
                                   int max_splits) {
    if (batch_nheads_mblocks >= 0.8f * num_sms) return 1;
    max_splits = imin(imin(max_splits, num_sms), num_n_blocks);
    float max_eff = 0.f;
    float eff[129];
    for (int s = 1; s <= max_splits; s++) {
        int eligible = s == 1 || ceil_div(num_n_blocks, s) != ceil_div(num_n_blocks, s - 1);
        if (!eligible) { eff[s] = 0.f; continue; }
        float n_waves = (float)(batch_nheads_mblocks * s) / num_sms;
        float e = n_waves / ceilf(n_waves);
        if (e > max_eff) max_eff = e;
        eff[s] = e;
    }
    for (int s = 1; s <= max_splits; s++) {
        int eligible = s == 1 || ceil_div(num_n_blocks, s) != ceil_div(num_n_blocks, s - 1);
        if (!eligible) continue;
        if (eff[s] >= 0.85f * max_eff) return s;
    }
    return 1;
}

/* csrc/flash_attn/src/block_info.h:15-34 */
typedef struct {
    int sum_s_q, sum_s_k, actual_seqlen_q, seqlen_k_cache, actual_seqlen_k;
} block_info;

static block_info make_block_info(const fa_oracle_params *p, int bidb) {
    block_info bi;
    bi.sum_s_q = p->cu_seqlens_q ? p->cu_seqlens_q[bidb] : -1;
    bi.sum_s_k = (p->cu_seqlens_k && p->is_seqlens_k_cumulative) ? p->cu_seqlens_k[bidb] : -1;
    bi.actual_seqlen_q = p->cu_seqlens_q ? p->cu_seqlens_q[bidb + 1] - bi.sum_s_q : p->seqlen_q;
    bi.seqlen_k_cache = !p->cu_seqlens_k
                            ? p->seqlen_k
                            : (p->is_seqlens_k_cumulative ? p->cu_seqlens_k[bidb + 1] - bi.sum_s_k
                                                          : p->cu_seqlens_k[bidb]);
    bi.actual_seqlen_k = p->seqused_k ? p->seqused_k[bidb]
                                      : bi.seqlen_k_cache + (p->knew ? p->seqlen_knew : 0);
    return bi;
}

/* Row pointer of key/value row `row` of batch bidb, kv head hk.
 * Dense/varlen: block_info.h:31-34; cache_batch_idx: flash_fwd_kernel.h:559;
 * paged: utils.h:297-315 (page = block_table[row / page_block_size]). */
static float *kv_row(const fa_oracle_params *p, float *base, const block_info *bi, int bidb,
                     int hk, int row) {
    const int64_t d = p->d, hkn = p->h_k;
    if (p->block_table) {
        const int *bt = p->block_table + (int64_t)bidb * p->block_table_stride;
        int page = bt[row / p->page_block_size];
        int in_page = row % p->page_block_size;
        return base + (((int64_t)page * p->page_block_size + in_page) * hkn + hk) * d;
    }
    if (bi->sum_s_k >= 0) return base + (((int64_t)bi->sum_s_k + row) * hkn + hk) * d;
    int bc = p->cache_batch_idx ? p->cache_batch_idx[bidb] : bidb;
    return base + ((((int64_t)bc * p->seqlen_k) + row) * hkn + hk) * d;
}

static const float *q_row(const fa_oracle_params *p, const float *base, const block_info *bi,
                          int bidb, int h, int row) {
    const int64_t d = p->d, hn = p->h;
    if (bi->sum_s_q >= 0) return base + (((int64_t)bi->sum_s_q + row) * hn + h) * d;
    return base + ((((int64_t)bidb * p->seqlen_q) + row) * hn + h) * d;
}

/* rotary.h:19-78 (interleaved) and :82-148 (contiguous): fp32 math on 16-bit inputs,
 * result rounded back to the 16-bit type.  `pos` selects the cos/sin row. */
static void rotary_row(const fa_oracle_params *p, const float *src, float *dst, int pos) {
    const int rd = p->rotary_dim, half = rd / 2;
    const float *c = p->rotary_cos + (int64_t)pos * half;
    const float *s = p->rotary_sin + (int64_t)pos * half;
    float tmp[MAXD];
    for (int i = 0; i < p->d; i++) tmp[i] = src[i];
    if (p->is_rotary_interleaved) {
        for (int i = 0; i < half; i++) {
            float x0 = src[2 * i], x1 = src[2 * i + 1];
            tmp[2 * i] = fa_oracle_round(x0 * c[i] - x1 * s[i], p->round_mode);
            tmp[2 * i + 1] = fa_oracle_round(x0 * s[i] + x1 * c[i], p->round_mode);
        }
    } else {
        for (int i = 0; i < half; i++) {
            float x0 = src[i], x1 = src[i + half];
            tmp[i] = fa_oracle_round(x0 * c[i] + x1 * (-s[i]), p->round_mode);
            tmp[i + half] = fa_oracle_round(x1 * c[i] + x0 * s[i], p->round_mode);
        }
    }
    for (int i = 0; i < p->d; i++) dst[i] = tmp[i];
}

/* flash_fwd_kernel.h:672-782: copy knew/vnew into the cache at [seqlen_k_cache, actual_seqlen_k),
 * K optionally rotated with position = its index in the sequence (:681). */
static void append_kv(fa_oracle_params *p) {
    for (int b = 0; b < p->b; b++) {
        block_info bi = make_block_info(p, b);
        for (int hk = 0; hk < p->h_k; hk++) {
            for (int t = 0; t < p->seqlen_knew; t++) {
                int row = bi.seqlen_k_cache + t;
                if (row >= bi.actual_seqlen_k) break;
                const float *ks = p->knew + ((((int64_t)b * p->seqlen_knew) + t) * p->h_k + hk) * p->d;
                const float *vs = p->vnew + ((((int64_t)b * p->seqlen_knew) + t) * p->h_k + hk) * p->d;
                float *kd = kv_row(p, p->k, &bi, b, hk, row);
                float *vd = kv_row(p, p->v, &bi, b, hk, row);
                memcpy(vd, vs, sizeof(float) * p->d);
                if (p->rotary_dim > 0) rotary_row(p, ks, kd, row);
                else memcpy(kd, ks, sizeof(float) * p->d);
            }
        }
    }
}

/* philox.cuh:22-29: one Philox-4x32 round. */
static void philox_round(uint32_t c[4], const uint32_t key[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0]; /* kPhiloxSA * ctr.x */
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2]; /* kPhiloxSB * ctr.z */
    const uint32_t r0 = (uint32_t)(p1 >> 32) ^ c[1] ^ key[0];
    const uint32_t r1 = (uint32_t)p1;
    const uint32_t r2 = (uint32_t)(p0 >> 32) ^ c[3] ^ key[1];
    const uint32_t r3 = (uint32_t)p0;
    c[0] = r0; c[1] = r1; c[2] = r2; c[3] = r3;
}

/* philox.cuh:31-49: counter = (offset, subsequence), key = seed; n_rounds - 1 rounds each followed
 * by a key bump, then one more round.  The reference runs 7 rounds in all. */
void fa_oracle_philox(uint64_t seed, uint64_t subsequence, uint64_t offset, int n_rounds,
                      uint32_t out[4]) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t c[4] = {(uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)subsequence,
                     (uint32_t)(subsequence >> 32)};
    for (int i = 0; i < n_rounds - 1; i++) {
        philox_round(c, key);
        key[0] += 0x9E3779B9u;
        key[1] += 0xBB67AE85u;
    }
    philox_round(c, key);
    for (int i = 0; i < 4; i++) out[i] = c[i];
}

/* Which random byte decides entry (row, col).  The reference draws one 128-bit Philox value per
 * thread per 16 x 32 block of the attention matrix (dropout.h:39-44: subsequence = (row / 16,
 * col / 32), offset + (bid * nheads + hid) * 32 + lane, dropout.h:21) and spends its 16 bytes on
 * the 16 entries the thread holds of that block in the m16n8k16 accumulator layout: lane t owns
 * rows t/4 and t/4 + 8, columns 2 (t%4) + {0, 1} of each of the four 8-column atoms; byte
 * j*8 + a*4 + v (dropout.h:46,59-63,79 with the (4,2) regrouping of utils.h:215-223) goes to atom
 * 2j + a, value v = 2 (row >= 8) + (col & 1).  An entry is kept iff byte <= keep_u8 (:69,79). */
int fa_oracle_dropout_keep(uint64_t seed, uint64_t offset, int bid, int hid, int nheads, int row,
                           int col, int keep_u8) {
    const int r16 = row & 15, c32 = col & 31;
    const int lane = (r16 & 7) * 4 + (c32 & 7) / 2;
    const int atom = c32 >> 3;
    const int v = 2 * (r16 >> 3) + (c32 & 1);
    const int byte_idx = (atom >> 1) * 8 + (atom & 1) * 4 + v;
    const uint64_t subseq = (uint64_t)(uint32_t)(row >> 4) | ((uint64_t)(uint32_t)(col >> 5) << 32);
    const uint64_t off = offset + (uint64_t)((bid * nheads + hid) * 32 + lane);
    uint32_t rnd[4];
    fa_oracle_philox(seed, subseq, off, 7, rnd);
    const int byte = (rnd[byte_idx >> 2] >> (8 * (byte_idx & 3))) & 0xff;
    return byte <= keep_u8;
}

/* flash_api.cpp:106,111 */
static int keep_threshold(float p_dropout) { return (int)(uint8_t)floor((1.f - p_dropout) * 255.0); }

void fa_oracle_dropout_mask(uint64_t seed, uint64_t offset, int b, int h, int seqlen_q,
                            int seqlen_k, float p_dropout, uint8_t *mask) {
    const int thr = keep_threshold(p_dropout);
#pragma omp parallel for collapse(2)
    for (int bb = 0; bb < b; bb++)
        for (int hh = 0; hh < h; hh++)
            for (int i = 0; i < seqlen_q; i++)
                for (int j = 0; j < seqlen_k; j++)
                    mask[(((size_t)bb * h + hh) * seqlen_q + i) * seqlen_k + j] =
                        (uint8_t)fa_oracle_dropout_keep(seed, offset, bb, hh, h, i, j, thr);
}

typedef struct {
    int is_causal, is_local, wl, wr;
    float scale, scale_log2;
} mask_cfg;

/* One (batch, head, m_block[, split]) tile pass: flash_fwd_kernel.h:27-470 (standard kernel) and
 * :474-1053 (split kernel: same loop restricted to [n_split_idx*n_blocks_per_split, ...)).
 * Writes normalised O rows and LSE rows into o_rows (bm x d, fp32, NOT yet rounded) / lse_rows.
 * Returns 0 if the tile early-exited with "no keys" (O = 0, LSE = +inf or -inf when split). */
static void tile_pass(const fa_oracle_params *p, const mask_cfg *mc, const block_info *bi,
                      int bidb, int bidh, int m_block, int BM, int BN, int split_idx,
                      int num_splits, int is_split, float *o_rows, float *lse_rows) {
    const int d = p->d;
    const int hk = bidh / (p->h / p->h_k);
    const int sq = bi->actual_seqlen_q, sk = bi->actual_seqlen_k;
    const int rows = imin(BM, sq - m_block * BM);
    int n_min, n_max;
    if (!is_split) {
        /* flash_fwd_kernel.h:59-67 */
        n_min = !mc->is_local ? 0 : imax(0, (m_block * BM + sk - sq - mc->wl) / BN);
        n_max = ceil_div(sk, BN);
        if (mc->is_causal || mc->is_local)
            n_max = imin(n_max, ceil_div((m_block + 1) * BM + sk - sq + mc->wr, BN));
    } else {
        /* flash_fwd_kernel.h:504-512: blocks per split from the CAPACITY params.seqlen_k */
        int per = ceil_div(ceil_div(p->seqlen_k, BN), num_splits);
        n_min = !mc->is_local ? split_idx * per
                              : imax(split_idx * per, (m_block * BM + sk - sq - mc->wl) / BN);
        n_max = imin(ceil_div(sk, BN), (split_idx + 1) * per);
        if (mc->is_causal || mc->is_local)
            n_max = imin(n_max, ceil_div((m_block + 1) * BM + sk - sq + mc->wr, BN));
    }
    const float empty_lse = (is_split && num_splits > 1) ? -INFINITY : INFINITY;
    if (n_max <= n_min) { /* flash_fwd_kernel.h:70-106 / :513-552 */
        for (int r = 0; r < rows; r++) {
            for (int c = 0; c < d; c++) o_rows[r * d + c] = 0.f;
            lse_rows[r] = empty_lse;
        }
        return;
    }
    float alibi = 0.f;
    if (p->alibi_slopes) /* flash_fwd_kernel.h:266: slope / scale_softmax, added pre-scale */
        alibi = p->alibi_slopes[(int64_t)bidb * p->alibi_batch_stride + bidh] / mc->scale;

    const int drop = p->p_dropout > 0.f;
    const int keep_u8 = keep_threshold(p->p_dropout);
    const float rp_dropout = 1.f / (1.f - p->p_dropout); /* flash_api.cpp:112 */
    float s[256];
    for (int r = 0; r < rows; r++) {
        const int row_idx = m_block * BM + r;
        const float *qr = q_row(p, p->q, bi, bidb, bidh, row_idx);
        float qrot[MAXD];
        if (p->knew && p->rotary_dim > 0) { /* flash_fwd_kernel.h:785-824 */
            int pos = bi->seqlen_k_cache + ((mc->is_causal || mc->is_local) ? row_idx : 0);
            if (p->seqlen_ro > 0 && pos > p->seqlen_ro - 1) pos = p->seqlen_ro - 1;
            rotary_row(p, qr, qrot, pos);
            qr = qrot;
        }
        float acc[MAXD];
        for (int c = 0; c < d; c++) acc[c] = 0.f;
        float row_max = -INFINITY, row_sum = 0.f;
        /* mask.h:171-172 */
        const int lim_left = imax(0, row_idx + sk - sq - mc->wl);
        const int lim_right = imin(sk, row_idx + 1 + sk - sq + mc->wr);
        int first = 1;
        for (int n = n_max - 1; n >= n_min; n--) { /* REVERSE: flash_fwd_kernel.h:246,281,354 */
            const int c0 = n * BN, cn = imin(BN, sk - c0);
            for (int j = 0; j < BN; j++) {
                const int col = c0 + j;
                if (j >= cn) { s[j] = -INFINITY; continue; } /* mask.h:159,199 */
                const float *kr = kv_row(p, p->k, bi, bidb, hk, col);
                float a = 0.f;
                for (int c = 0; c < d; c++) a += qr[c] * kr[c]; /* flash_fwd_kernel.h:298-301 */
                if (p->alibi_slopes) { /* mask.h:156,179-186 */
                    if (mc->is_causal) a += alibi * (float)col;
                    else a -= alibi * fabsf((float)(row_idx + sk - sq - col));
                }
                if (mc->is_causal && col >= lim_right) a = -INFINITY;          /* mask.h:187-191 */
                if (mc->is_local && (col >= lim_right || col < lim_left)) a = -INFINITY; /* :192-196 */
                s[j] = a;
            }
            /* softmax.h:135-166 */
            float m_prev = row_max, m_new = row_max;
            for (int j = 0; j < BN; j++) m_new = fmaxf(m_new, s[j]);
            const float m_cur = (m_new == -INFINITY) ? 0.f : m_new;
            if (!first) {
                const float sc = exp2f((m_prev - m_cur) * mc->scale_log2); /* softmax.h:156 */
                row_sum *= sc;
                for (int c = 0; c < d; c++) acc[c] *= sc;
            }
            row_max = m_new;
            const float max_scaled = m_cur * mc->scale_log2; /* softmax.h:75 */
            for (int j = 0; j < BN; j++) {
                float pj = exp2f(s[j] * mc->scale_log2 - max_scaled); /* softmax.h:87 */
                row_sum += pj;                                        /* fp32 P: softmax.h:164 */
                s[j] = fa_oracle_round(pj, p->round_mode);            /* flash_fwd_kernel.h:323 */
            }
            if (drop || p->s_dmask) { /* flash_fwd_kernel.h:324-337: on the ROUNDED P, after the row sum */
                for (int j = 0; j < BN; j++) {
                    const int keep = !drop || fa_oracle_dropout_keep(p->philox_seed, p->philox_offset,
                                                                     bidb, bidh, p->h, row_idx, c0 + j, keep_u8);
                    if (p->s_dmask && j < cn) /* sign bit = dropped (dropout.h:31-33) */
                        p->s_dmask[(((int64_t)bidb * p->h + bidh) * p->seqlen_q_rounded + row_idx) *
                                       p->seqlen_k_rounded + c0 + j] = keep ? s[j] : -s[j];
                    if (!keep) s[j] = 0.f;
                }
            }
            for (int j = 0; j < cn; j++) { /* V tail rows are zero-filled: kernel.h:291-294 */
                if (s[j] == 0.f) continue;
                const float *vr = kv_row(p, p->v, bi, bidb, hk, c0 + j);
                for (int c = 0; c < d; c++) acc[c] += s[j] * vr[c]; /* flash_fwd_kernel.h:343 */
            }
            first = 0;
        }
        /* softmax.h:168-185 */
        const int bad = (row_sum == 0.f || row_sum != row_sum);
        const float inv_sum = bad ? 1.f : 1.f / row_sum;
        const float inv = drop ? inv_sum * rp_dropout : inv_sum; /* softmax.h:181 */
        lse_rows[r] = bad ? empty_lse : row_max * mc->scale + logf(row_sum);
        for (int c = 0; c < d; c++) o_rows[r * d + c] = acc[c] * inv;
    }
}

int fa_oracle_fwd(fa_oracle_params *p) {
    if (p->d > MAXD || p->d <= 0 || p->h % p->h_k != 0) return -1;
    if (p->knew) append_kv(p);

    mask_cfg mc;
    int wl = p->window_left, wr = p->window_right, causal = p->is_causal, local = 0;
    fa_oracle_fold_window(p->seqlen_q, p->seqlen_k, p->alibi_slopes != NULL, &causal, &wl, &wr,
                          &local);
    mc.is_causal = causal; mc.is_local = local; mc.wl = wl; mc.wr = wr;
    mc.scale = p->softmax_scale;
    mc.scale_log2 = p->softmax_scale * LOG2E; /* flash_api.cpp:103 */

    const int d = p->d;
    const int split = p->num_splits > 1;
    int BM, BN;
    if (!split) { /* flash_fwd_launch_template.h:168-370 (tile per head dim) */
        BM = p->block_m > 0 ? p->block_m : 128;
        BN = p->block_n > 0 ? p->block_n : (d <= 64 ? 128 : 64);
    } else {      /* flash_fwd_launch_template.h:160-165 */
        BM = p->block_m > 0 ? p->block_m : 64;
        BN = p->block_n > 0 ? p->block_n : (d <= 64 ? 256 : (d <= 128 ? 128 : 64));
    }
    if (BN > 256) return -2;
    const int nsplit = split ? p->num_splits : 1;

    /* seqlen_k == 0: flash_api.cpp:401-405 */
    int rc = 0;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < p->b; b++) {
        for (int h = 0; h < p->h; h++) {
            block_info bi = make_block_info(p, b);
            const int sq = bi.actual_seqlen_q;
            const int n_mblocks = ceil_div(imax(sq, 0), BM);
            float *o_s = (float *)malloc(sizeof(float) * (size_t)nsplit * BM * d);
            float *l_s = (float *)malloc(sizeof(float) * (size_t)nsplit * BM);
            for (int mb = 0; mb < n_mblocks; mb++) {
                const int rows = imin(BM, sq - mb * BM);
                for (int sp = 0; sp < nsplit; sp++)
                    tile_pass(p, &mc, &bi, b, h, mb, BM, BN, sp, nsplit, split,
                              o_s + (size_t)sp * BM * d, l_s + (size_t)sp * BM);
                for (int r = 0; r < rows; r++) {
                    const int row_idx = mb * BM + r;
                    float *orow = (float *)q_row(p, p->out, &bi, b, h, row_idx);
                    float *lse = p->lse + ((int64_t)b * p->h + h) * p->seqlen_q + row_idx;
                    if (!split) {
                        for (int c = 0; c < d; c++)
                            orow[c] = fa_oracle_round(o_s[r * d + c], p->round_mode);
                        *lse = l_s[r];
                    } else {
                        /* combine: flash_fwd_kernel.h:1150-1221 */
                        float lmax = -INFINITY;
                        for (int sp = 0; sp < nsplit; sp++) lmax = fmaxf(lmax, l_s[sp * BM + r]);
                        if (lmax == -INFINITY) lmax = 0.f;
                        float lsum = 0.f;
                        for (int sp = 0; sp < nsplit; sp++) lsum += expf(l_s[sp * BM + r] - lmax);
                        float ltot = (lsum == 0.f || lsum != lsum) ? INFINITY : logf(lsum) + lmax;
                        *lse = ltot;
                        for (int c = 0; c < d; c++) {
                            float a = 0.f;
                            for (int sp = 0; sp < nsplit; sp++)
                                a += expf(l_s[sp * BM + r] - ltot) * o_s[((size_t)sp * BM + r) * d + c];
                            orow[c] = fa_oracle_round(a, p->round_mode);
                        }
                    }
                }
            }
            free(o_s);
            free(l_s);
        }
    }
    return rc;
}

/* Backward.  The reference tree has no backward kernel (setup.py:154-169, flash_api.cpp:943-944);
 * its boundary is pinned by vllm_flash_attn/flash_attn_interface.py:135-154 and the parameter
 * block csrc/flash_attn/src/flash.h:145-183.  Math: P = exp2(S*scale_log2 - LSE*log2e) with the
 * forward's mask/bias, dV += P16^T dO, dP = dO V^T, delta = rowsum(dO*O), dS = P*(dP - delta),
 * dQ += scale * dS16 K, dK += scale * dS16^T Q; dK/dV of a KV head sum over its query heads.
 * With dropout (keep mask M, keep probability pk, the forward's Philox stream): dV uses P16 * M,
 * delta is scaled by pk, dS = P * (M * dP - pk * delta), and dQ, dK (dV) carry the factor
 * scale / pk (1 / pk) at the end -- the scale_softmax_rp_dropout / rp_dropout fields of
 * flash.h:109-117 (the upstream FlashAttention-2 backward formulation). */
int fa_oracle_bwd(fa_oracle_params *p) {
    if (p->d > MAXD || p->h % p->h_k != 0) return -1;
    mask_cfg mc;
    int wl = p->window_left, wr = p->window_right, causal = p->is_causal, local = 0;
    fa_oracle_fold_window(p->seqlen_q, p->seqlen_k, p->alibi_slopes != NULL, &causal, &wl, &wr,
                          &local);
    mc.is_causal = causal; mc.is_local = local; mc.wl = wl; mc.wr = wr;
    mc.scale = p->softmax_scale;
    mc.scale_log2 = p->softmax_scale * LOG2E;
    const int d = p->d, g = p->h / p->h_k;
    const int drop = p->p_dropout > 0.f;
    const int keep_u8 = keep_threshold(p->p_dropout);
    const float p_keep = 1.f - p->p_dropout;
    const float rp_dropout = 1.f / p_keep;
    const float scale_rp = drop ? rp_dropout * mc.scale : mc.scale; /* flash_api.cpp:113 */

#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < p->b; b++) {
        for (int hk = 0; hk < p->h_k; hk++) {
            block_info bi = make_block_info(p, b);
            const int sq = bi.actual_seqlen_q, sk = bi.actual_seqlen_k;
            float *dkacc = (float *)calloc((size_t)imax(sk, 1) * d, sizeof(float));
            float *dvacc = (float *)calloc((size_t)imax(sk, 1) * d, sizeof(float));
            for (int hh = 0; hh < g; hh++) {
                const int h = hk * g + hh;
                float alibi = 0.f;
                if (p->alibi_slopes)
                    alibi = p->alibi_slopes[(int64_t)b * p->alibi_batch_stride + h] / mc.scale;
                for (int i = 0; i < sq; i++) {
                    const float *qr = q_row(p, p->q, &bi, b, h, i);
                    const float *orow = q_row(p, p->out, &bi, b, h, i);
                    const float *dor = q_row(p, p->dout, &bi, b, h, i);
                    float *dqr = (float *)q_row(p, p->dq, &bi, b, h, i);
                    const float lse = p->lse[((int64_t)b * p->h + h) * p->seqlen_q + i];
                    float delta = 0.f;
                    for (int c = 0; c < d; c++) delta += dor[c] * orow[c];
                    if (drop) delta *= p_keep;
                    if (p->softmax_d)
                        p->softmax_d[((int64_t)b * p->h + h) * p->seqlen_q + i] = delta;
                    float dqa[MAXD];
                    for (int c = 0; c < d; c++) dqa[c] = 0.f;
                    const int lim_left = imax(0, i + sk - sq - mc.wl);
                    const int lim_right = imin(sk, i + 1 + sk - sq + mc.wr);
                    const int lo = mc.is_local ? lim_left : 0;
                    const int hi = (mc.is_causal || mc.is_local) ? lim_right : sk;
                    if (!(lse == INFINITY) && lse == lse) {
                        for (int j = lo; j < hi; j++) {
                            const float *kr = kv_row(p, p->k, &bi, b, hk, j);
                            const float *vr = kv_row(p, p->v, &bi, b, hk, j);
                            float a = 0.f, dp = 0.f;
                            for (int c = 0; c < d; c++) { a += qr[c] * kr[c]; dp += dor[c] * vr[c]; }
                            if (p->alibi_slopes) {
                                if (mc.is_causal) a += alibi * (float)j;
                                else a -= alibi * fabsf((float)(i + sk - sq - j));
                            }
                            float pj = exp2f(a * mc.scale_log2 - lse * LOG2E);
                            const int keep = !drop || fa_oracle_dropout_keep(p->philox_seed, p->philox_offset,
                                                                             b, h, p->h, i, j, keep_u8);
                            float ds = pj * ((keep ? dp : 0.f) - delta);
                            float p16 = keep ? fa_oracle_round(pj, p->round_mode) : 0.f;
                            float ds16 = fa_oracle_round(ds, p->round_mode);
                            float *dkr = dkacc + (size_t)j * d, *dvr = dvacc + (size_t)j * d;
                            for (int c = 0; c < d; c++) {
                                dvr[c] += p16 * dor[c];
                                dkr[c] += ds16 * qr[c];
                                dqa[c] += ds16 * kr[c];
                            }
                        }
                    }
                    for (int c = 0; c < d; c++)
                        dqr[c] = fa_oracle_round(dqa[c] * scale_rp, p->round_mode);
                }
            }
            for (int j = 0; j < sk; j++) {
                float *dkr = kv_row(p, p->dk, &bi, b, hk, j);
                float *dvr = kv_row(p, p->dv, &bi, b, hk, j);
                for (int c = 0; c < d; c++) {
                    dkr[c] = fa_oracle_round(dkacc[(size_t)j * d + c] * scale_rp, p->round_mode);
                    dvr[c] = fa_oracle_round(drop ? dvacc[(size_t)j * d + c] * rp_dropout
                                                  : dvacc[(size_t)j * d + c], p->round_mode);
                }
            }
            free(dkacc);
            free(dvacc);
        }
    }
    return 0;
}
