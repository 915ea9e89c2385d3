// Decode / split-KV forward, LSE-weighted combine, and KV-cache append (+ rotary) for gfx950.
//
// Replaces compute_attn_1rowblock_splitkv / flash_fwd_splitkv_kernel
// (csrc/flash_attn/src/flash_fwd_kernel.h:474-1053), combine_attn_seqk_parallel /
// flash_fwd_splitkv_combine_kernel (:1092-1250) and the Append_KV prologue (:672-824) of the
// reference, launched as in flash_fwd_launch_template.h:96-166.
//
// Decode is HBM-bound (a few query rows against a long K/V stream), so the tiling is turned
// around relative to the prefill kernel: a workgroup of 4 waves owns ONE block of up to 32 query
// rows and the waves split the KEYS.  Each iteration stages 128 keys of K and V (coalesced 16-byte
// loads -> VGPR -> LDS, next tile's loads in flight during the MFMAs); wave w computes keys
// [32 w, 32 w + 32) of the tile with the same swapped-product MFMA scheme as the prefill kernel
// and keeps its own running (max, sum, O); the four partial states are merged through LDS at the
// end.  With num_splits > 1 the key range is additionally cut across workgroups, each writing a
// normalised fp32 partial O and its LSE; a second kernel merges them.
#include "vfa_kernel_params.h"

namespace vfa {

constexpr int DEC_NW = 4;
constexpr int DEC_NT = DEC_NW * 64;
constexpr int DEC_BM = 32;
constexpr int DEC_BN = 32 * DEC_NW;  // keys per iteration


// Rotary embedding of one 8-element chunk [d0, d0+8) of a row: fp32 math on 16-bit inputs, result
// rounded back (rotary.h:57-63 interleaved pairs (2i, 2i+1); :116-132 contiguous pairs
// (i, i + rotary_dim/2)).  cosp/sinp point at the (rotary_dim/2)-wide row of the position.
template <bool BF16>
__device__ __forceinline__ i32x4 rotary_chunk(const unsigned short *src, int d0,
                                              const unsigned short *cosp, const unsigned short *sinp,
                                              int rotary_dim, bool interleaved) {
    if (rotary_dim == 0 || d0 >= rotary_dim) return *(const i32x4 *)(src + d0);
    const int half = rotary_dim / 2;
    unsigned short outv[8];
    if (interleaved) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float x0 = to_f32<BF16>(src[d0 + 2 * i]), x1 = to_f32<BF16>(src[d0 + 2 * i + 1]);
            const float c = to_f32<BF16>(cosp[d0 / 2 + i]), s = to_f32<BF16>(sinp[d0 / 2 + i]);
            const int pk = pack2<BF16>(x0 * c - x1 * s, x0 * s + x1 * c);
            outv[2 * i] = (unsigned short)(pk & 0xffff);
            outv[2 * i + 1] = (unsigned short)((unsigned)pk >> 16);
        }
    } else {
        const bool left = d0 < half;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int di = d0 + i;
            const float x = to_f32<BF16>(src[di]);
            const float xo = to_f32<BF16>(src[left ? di + half : di - half]);
            const int ci = left ? di : di - half;
            const float c = to_f32<BF16>(cosp[ci]), s = to_f32<BF16>(sinp[ci]);
            const float y = x * c + xo * (left ? -s : s);
            outv[i] = (unsigned short)(pack2<BF16>(y, 0.f) & 0xffff);
        }
    }
    i32x4 w;
    w[0] = outv[0] | (outv[1] << 16); w[1] = outv[2] | (outv[3] << 16);
    w[2] = outv[4] | (outv[5] << 16); w[3] = outv[6] | (outv[7] << 16);
    return w;
}

// Geometry of the decode kernel.  The basic workgroup is 4 waves that split the 128 keys of a staged
// tile between them.  Two cases double it to 8 waves (two groups of 4, one workgroup per CU):
//   * d > 128 (DSPLIT = 2): a tile's staging registers (256 bytes of K and of V per thread at 256
//     threads) plus the O accumulators do not fit 256 registers -- the kernel spilled to scratch and
//     ran at 2.6 TB/s.  Group g takes half g of the head dim of O; both groups compute the scores
//     of their 32 keys (duplicated MFMAs cost nothing on an HBM-bound kernel); every thread stages
//     half as many bytes; the query fragments are re-read from LDS instead of held in 64 registers.
//     d = 256 then reads K+V at 6.0-6.26 TB/s.
//   * HPB = 2 heads per workgroup at d = 128: group g takes head 2 j + g.  In the (batch, key, head,
//     d) layout the rows of adjacent heads are adjacent in memory, so the workgroup reads 512
//     contiguous bytes per key instead of 256.  b128 kv8192 hk8 d128: 5.71 -> 6.17 TB/s (a
//     read-only probe of 256-byte pieces at a 2 KiB stride against one contiguous stream per
//     workgroup shows 3 % in favour of longer pieces).  Four heads per workgroup at d = 64 (16
//     waves, 128 registers each) spilled and ran at half the rate of one head per workgroup.
template <int DP, int HPB>
struct DecodeCfg {
    static constexpr int CPR = DP / 8;         // 16-byte chunks of one head's row
    static constexpr int DSPLIT = DP > 128 ? 2 : 1;
    static constexpr int NGRP = DSPLIT * HPB;  // groups of DEC_NW waves
    static_assert(NGRP <= 2, "at most two groups of waves");
    static constexpr int NT = DEC_NT * NGRP;
    static constexpr int ROWE = DP * HPB;      // elements of one staged row (all heads of the workgroup)
    static constexpr int CPRR = ROWE / 8;
    static constexpr int TILE_CHUNKS = DEC_BN * CPRR;
    static constexpr int CH = TILE_CHUNKS / NT;
    static constexpr int TILE_BYTES = DEC_BN * ROWE * 2;
    static constexpr int KV_LDS = 2 * TILE_BYTES;
    static constexpr int O_ROWB = DP * 4 + 16;
    static constexpr int O_HEAD = DEC_NW * DEC_BM * O_ROWB + DEC_NW * DEC_BM * 8;  // merge area of one head
    static constexpr int O_LDS = HPB * O_HEAD;
    static constexpr int Q_LDS = DP > 128 ? DEC_BM * DP * 2 : 0;
    static constexpr int LDS_BYTES = KV_LDS + Q_LDS > O_LDS ? KV_LDS + Q_LDS : O_LDS;
    static constexpr int MIN_WG = NGRP >= 2 ? 1 : 2;
};

template <int DP, bool BF16, int HPB>
__global__ void __launch_bounds__((DecodeCfg<DP, HPB>::NT), (DecodeCfg<DP, HPB>::MIN_WG))
fwd_decode_kernel(const KParams p, const int n_mblocks) {
    using Cfg = DecodeCfg<DP, HPB>;
    constexpr int CPR = Cfg::CPR, CPRR = Cfg::CPRR, CH = Cfg::CH, NT = Cfg::NT, ROWE = Cfg::ROWE;
    constexpr int KS = DP / 16, NDB = DP / 32 / Cfg::DSPLIT;  // NDB: 32-wide blocks of O this wave accumulates
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_id & (DEC_NW - 1);   // which 32 keys of the staged tile
    const int grp = wave_id / DEC_NW;          // group of waves
    const int dhalf = Cfg::DSPLIT == 2 ? grp : 0;  // which part of the head dim of O
    const int hsel = HPB > 1 ? grp : 0;            // which head of the workgroup
    const int r = lane & 31, hh = lane >> 5;

    // block -> (split, batch, head block, m_block); consecutive blocks walk the heads of one batch.
    // Hardware deals consecutive block ids round-robin over the 8 XCDs.  With one split that puts
    // head x of every batch entry on XCD x, which measured best (b128 kv8192 hk8: 747 us).  With
    // several splits the ids are regrouped so that each XCD owns one contiguous range of the index,
    // i.e. all heads and splits of a batch entry run on one XCD at about the same time: 727 us
    // against 801 us ungrouped at 4 splits, 737 against 783 at 2.  Likewise with several row blocks
    // per head, which all read the head's K/V: they share one XCD's L2.
    int L = blockIdx.x;
    if ((p.num_splits > 1 || n_mblocks > 1) != ((p.dbg_flags & 1) != 0)) {
        const int per_xcd = gridDim.x / 8;
        if (L < per_xcd * 8) L = (L % 8) * per_xcd + L / 8;
    }
    // With pack_gqa the grid walks the KV heads and the rows of a workgroup are the (query position,
    // query head of the group) pairs of its KV head, row i = position i / ratio, head i % ratio: the
    // query heads that share a KV head read its K/V once instead of once each (one, two, ... query
    // tokens per sequence against a long cache: speculative decoding, small chunks, ALiBi decode).
    const int ratio = p.pack_gqa ? p.h_ratio : 1;
    const int n_hblocks = (p.pack_gqa ? p.h_k : p.h) / HPB;
    const int m_block = L % n_mblocks; L /= n_mblocks;
    const int bidh0 = (L % n_hblocks) * HPB; L /= n_hblocks;
    const int split = L % p.num_splits; L /= p.num_splits;
    const int bidb = L;
    const int vh = bidh0 + hsel;           // this wave's head: a KV head when packed, else a query head
    const int hk0 = p.pack_gqa ? bidh0 : bidh0 / p.h_ratio;  // first KV head of the workgroup
    const bool is_split = p.num_splits > 1;

    const BlockInfo bi(p, bidb);
    const int sq = bi.sq, sk = bi.sk;
    const int sqv = sq * ratio;            // rows of this (batch, head)
    if (m_block * DEC_BM >= sqv) return;
    const int q0 = m_block * DEC_BM;
    const int pos_lo = q0 / ratio, pos_hi = (q0 + DEC_BM - 1) / ratio;  // query positions of the block's rows

    // key tile range of this split (flash_fwd_kernel.h:504-512), in units of DEC_BN keys; the
    // per-split block count comes from the capacity p.seqlen_k like the reference's.
    const int per = (((p.seqlen_k + DEC_BN - 1) / DEC_BN) + p.num_splits - 1) / p.num_splits;
    int n_min = split * per;
    if (p.is_local) n_min = max(n_min, (pos_lo + sk - sq - p.wl) / DEC_BN);
    int n_max = min((sk + DEC_BN - 1) / DEC_BN, (split + 1) * per);
    if (p.is_causal || p.is_local)
        n_max = min(n_max, (pos_hi + 1 + sk - sq + p.wr + DEC_BN - 1) / DEC_BN);

    const unsigned short *qg = (const unsigned short *)p.q + bi.q_off(p.q_bs, p.q_rs, bidb);
    unsigned short *og = (unsigned short *)p.o + bi.q_off(p.o_bs, p.o_rs, bidb);
    // row i of this wave's head -> query position and query head, and where its results go
    auto row_pos = [&](int i) { return i / ratio; };
    auto row_head = [&](int i) { return vh * ratio + (i - (i / ratio) * ratio); };
    auto lse_ptr = [&](int i) -> float * {
        const int64_t bh = (int64_t)bidb * p.h + row_head(i);
        return is_split ? p.lseaccum + ((int64_t)split * p.b * p.h + bh) * p.seqlen_q + row_pos(i)
                        : p.lse + bh * p.seqlen_q + row_pos(i);
    };
    auto oacc_ptr = [&](int i) -> float * {
        const int64_t bh = (int64_t)bidb * p.h + row_head(i);
        return p.oaccum + (((int64_t)split * p.b * p.h + bh) * p.seqlen_q + row_pos(i)) * p.d;
    };
    auto o_ptr = [&](int i) -> unsigned short * {
        return og + (int64_t)row_pos(i) * p.o_rs + (int64_t)row_head(i) * p.o_hs;
    };
    // threads that share this wave's head: the whole workgroup, or its group of 4 waves
    const int tid_h = HPB > 1 ? (tid & (DEC_NT - 1)) : tid;
    constexpr int NT_H = HPB > 1 ? DEC_NT : NT;

    if (n_max <= n_min) {
        // flash_fwd_kernel.h:513-552: O = 0 and LSE = -inf (split) / +inf (final)
        for (int c = tid_h; c < DEC_BM * CPR; c += NT_H) {
            const int row = q0 + c / CPR, ch = c % CPR;
            if (row < sqv && ch * 8 < p.d) {
                if (is_split) {
                    float *dst = oacc_ptr(row) + ch * 8;
                    *(f32x4 *)dst = f32x4{0, 0, 0, 0};
                    *(f32x4 *)(dst + 4) = f32x4{0, 0, 0, 0};
                } else {
                    *(i32x4 *)(o_ptr(row) + ch * 8) = i32x4{0, 0, 0, 0};
                }
            }
        }
        for (int c = tid_h; c < DEC_BM; c += NT_H)
            if (q0 + c < sqv) *lse_ptr(q0 + c) = is_split ? -INFINITY : INFINITY;
        return;
    }

    const int bidb_cache = p.cache_batch_idx ? p.cache_batch_idx[bidb] : bidb;
    const unsigned short *kg = (const unsigned short *)p.k + (int64_t)hk0 * p.k_hs;
    const unsigned short *vg = (const unsigned short *)p.v + (int64_t)hk0 * p.v_hs;
    const int *bt = nullptr;
    if (p.block_table) {
        bt = p.block_table + (int64_t)bidb * p.bt_bs;
    } else {
        kg += bi.k_off(p.k_bs, p.k_rs, bidb_cache);
        vg += bi.k_off(p.v_bs, p.v_rs, bidb_cache);
    }

    // chunk c of the tile: key c / CPRR, and within the staged row head (c % CPRR) / CPR, 16-byte
    // piece (c % CPRR) % CPR -- consecutive threads read consecutive 16 bytes of a head's row and
    // then of the next head's
    i32x4 kreg[CH], vreg[CH];
    // Paged KV: the page ids of a tile are fetched one tile ahead of its rows, so that the
    // block-table lookup is not a second memory latency in front of every K/V load (decode is
    // HBM-bound and prefetches only one tile deep).
    int pg[CH];
    auto load_pages = [&](int n) {
        if (!bt) return;
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int grow = n * DEC_BN + (tid + i * NT) / CPRR;
            const int safe = grow < sk ? grow : 0;
            pg[i] = bt[p.page_shift >= 0 ? safe >> p.page_shift : safe / p.page_size];
        }
    };
    auto load_tile = [&](int n) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int c = tid + i * NT;
            const int row = c / CPRR, chr = c % CPRR;
            const int hd = chr / CPR, ch = chr % CPR;
            const int grow = n * DEC_BN + row;
            const bool ok = grow < sk && ch * 8 < p.d;
            int64_t ko, vo;
            if (bt) {  // csrc/flash_attn/src/utils.h:297-315
                const int safe = ok ? grow : 0;
                const int inpage = p.page_shift >= 0 ? safe & (p.page_size - 1) : safe % p.page_size;
                ko = (int64_t)pg[i] * p.k_bs + (int64_t)inpage * p.k_rs;
                vo = (int64_t)pg[i] * p.v_bs + (int64_t)inpage * p.v_rs;
            } else {
                ko = (int64_t)grow * p.k_rs;
                vo = (int64_t)grow * p.v_rs;
            }
            if constexpr (HPB > 1) { ko += (int64_t)hd * p.k_hs; vo += (int64_t)hd * p.v_hs; }
            kreg[i] = ok ? *(const i32x4 *)(kg + ko + ch * 8) : i32x4{0, 0, 0, 0};
            vreg[i] = ok ? *(const i32x4 *)(vg + vo + ch * 8) : i32x4{0, 0, 0, 0};
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int c = tid + i * NT;
            const int off = lds_off<ROWE>(c / CPRR, c % CPRR);
            *(i32x4 *)(smem + off) = kreg[i];
            *(i32x4 *)(smem + Cfg::TILE_BYTES + off) = vreg[i];
        }
    };

    // Q fragments: every wave (of a head) holds the same 32 query rows
    i32x4 qf[KS];
    {
        const int row = q0 + r;
        const bool rok = row < sqv;
        const int rsafe = rok ? row : q0;
        const unsigned short *qr = qg + (int64_t)row_pos(rsafe) * p.q_rs + (int64_t)row_head(rsafe) * p.q_hs;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int col = 16 * s + 8 * hh;
            qf[s] = (rok && col < p.d) ? *(const i32x4 *)(qr + col) : i32x4{0, 0, 0, 0};
        }
        if (p.knew && p.rotary_dim > 0 && rok) {
            // flash_fwd_kernel.h:785-824: the query is rotated with position seqlen_k_cache
            // (+ its own position when causal or local)
            int pos = bi.sk_cache + ((p.is_causal || p.is_local) ? row_pos(row) : 0);
            if (p.seqlen_ro > 0) pos = min(pos, p.seqlen_ro - 1);  // never past the table's end
            const int half = p.rotary_dim / 2;
            const unsigned short *cosp = (const unsigned short *)p.rotary_cos + (int64_t)pos * half;
            const unsigned short *sinp = (const unsigned short *)p.rotary_sin + (int64_t)pos * half;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const int col = 16 * s + 8 * hh;
                if (col < p.d)
                    qf[s] = rotary_chunk<BF16>(qr, col, cosp, sinp, p.rotary_dim, p.rotary_interleaved);
            }
        }
    }
    // The first tile is requested before the query fragments are touched: wave 0's copy of them
    // into LDS waits for their loads, and the page ids and K/V rows behind that wait would make
    // the prologue one memory round trip longer (a sequence's first tile is its whole life at
    // small batch with many splits).
    load_pages(n_max - 1);
    load_tile(n_max - 1);
    if (n_max - 1 > n_min) load_pages(n_max - 2);

    char *q_lds = smem + Cfg::KV_LDS;  // fragment s of lane l at (s * 64 + l) * 16
    if constexpr (Cfg::Q_LDS > 0) {
        if (wave_id == 0) {
#pragma unroll
            for (int s = 0; s < KS; s++) *(i32x4 *)(q_lds + (s * 64 + lane) * 16) = qf[s];
        }
        // (the first tile's barrier below orders these stores before any wave reads them)
    }

    f32x16 o[NDB];
#pragma unroll
    for (int i = 0; i < NDB; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) o[i][j] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    float alibi = 0.f;
    const bool has_alibi = p.alibi_slopes != nullptr;
    const int my_pos = row_pos(q0 + r);  // query position of this lane's row
    if (has_alibi) alibi = p.alibi_slopes[(int64_t)bidb * p.alibi_bs + row_head(min(q0 + r, sqv - 1))] / p.scale;
    const float c2 = p.scale_log2;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;

    const char *kb = smem;
    const char *vb = smem + Cfg::TILE_BYTES;
    const int wrow = 32 * wave;      // this wave's rows of the staged tile
    const int hch = hsel * CPR;      // first 16-byte chunk of this wave's head in a staged row

    for (int n = n_max - 1; n >= n_min; n--) {
        store_tile();
        __syncthreads();
        if (n > n_min) {
            load_tile(n - 1);
            if (n - 1 > n_min) load_pages(n - 2);
        }

        f32x16 s;
#pragma unroll
        for (int j = 0; j < 16; j++) s[j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const i32x4 kf = *(const i32x4 *)(kb + lds_off<ROWE>(wrow + r, hch + 2 * ks + hh));
            if constexpr (Cfg::Q_LDS > 0) s = mma32<BF16>(kf, *(const i32x4 *)(q_lds + (ks * 64 + lane) * 16), s);
            else s = mma32<BF16>(kf, qf[ks], s);
        }

        const int c0 = n * DEC_BN + wrow;
        bool need = has_alibi || (c0 + 32 > sk);
        if (p.is_causal || p.is_local) need = need || (c0 + 32 > pos_lo + 1 + sk - sq + p.wr);
        if (p.is_local) need = need || (c0 < pos_hi + sk - sq - p.wl);
        if (need) {
            const int row = my_pos;
            const int lo = p.is_local ? max(0, row + sk - sq - p.wl) : 0;
            const int hi = (p.is_causal || p.is_local) ? min(sk, row + 1 + sk - sq + p.wr) : sk;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int col = c0 + (j & 3) + 8 * (j >> 2) + 4 * hh;
                float x = s[j];
                if (has_alibi)
                    x = p.is_causal ? x + alibi * (float)col
                                    : x - alibi * fabsf((float)(row + sk - sq - col));
                s[j] = (col >= lo && col < hi) ? x : -INFINITY;
            }
        }

        float mx = s[0];
#pragma unroll
        for (int j = 1; j < 16; j++) mx = fmaxf(mx, s[j]);
        mx = half_swap_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float m_cur = (m_new == -INFINITY) ? 0.f : m_new;
        if (!__all(m_new == m_run)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_cur) * c2);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < NDB; i++)
#pragma unroll
                for (int j = 0; j < 16; j++) o[i][j] *= alpha;
        }
        m_run = m_new;
        const float mc = m_cur * c2;
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j], c2, -mc));
            s[j] = e;
            psum += e;
        }
        l_run += psum;

        i32x4 pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int w = 0; w < 4; w++)
                pf[ks][w] = pack2<BF16>(s[8 * ks + 2 * w], s[8 * ks + 2 * w + 1]);
#pragma unroll
        for (int db = 0; db < NDB; db++) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const int row0 = wrow + 16 * ks + 4 * hh + tr_q;
                const int ch = hch + 4 * (db + dhalf * NDB) + 2 * tr_g + (tr_p >> 1);
                const s16x4 v0 = lds_read_tr16(vb + lds_off<ROWE>(row0, ch) + 8 * (tr_p & 1));
                const s16x4 v1 = lds_read_tr16(vb + lds_off<ROWE>(row0 + 8, ch) + 8 * (tr_p & 1));
                i32x4 vf;
                const i32x2 a0 = __builtin_bit_cast(i32x2, v0), a1 = __builtin_bit_cast(i32x2, v1);
                vf[0] = a0[0]; vf[1] = a0[1]; vf[2] = a1[0]; vf[3] = a1[1];
                o[db] = mma32<BF16>(vf, pf[ks], o[db]);
            }
        }
        __syncthreads();  // everyone is done with the staged tile before it is overwritten
    }

    // ---- merge the four waves' partial states of a head (same algebra as the split combine) ----
    char *area = smem + hsel * Cfg::O_HEAD;
    float *ml = (float *)(area + DEC_NW * DEC_BM * Cfg::O_ROWB);  // [wave][row][2]
    const float l_w = half_swap_sum(l_run);
    if (hh == 0 && dhalf == 0) {
        ml[(wave * DEC_BM + r) * 2 + 0] = m_run;
        ml[(wave * DEC_BM + r) * 2 + 1] = l_w;
    }
    __syncthreads();
    float m_all = -INFINITY;
#pragma unroll
    for (int w = 0; w < DEC_NW; w++) m_all = fmaxf(m_all, ml[(w * DEC_BM + r) * 2]);
    const float m_fin = (m_all == -INFINITY) ? 0.f : m_all;
    float l_all = 0.f;
#pragma unroll
    for (int w = 0; w < DEC_NW; w++)
        l_all += ml[(w * DEC_BM + r) * 2 + 1] *
                 __builtin_amdgcn_exp2f((ml[(w * DEC_BM + r) * 2] - m_fin) * c2);
    const float my_scale = __builtin_amdgcn_exp2f((m_run - m_fin) * c2);
    const bool bad = (l_all == 0.f) || (l_all != l_all);
    const float inv = bad ? 1.f : 1.f / l_all;
    const float f = my_scale * inv;
    char *ob = area + wave * DEC_BM * Cfg::O_ROWB;
#pragma unroll
    for (int db = 0; db < NDB; db++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            f32x4 w4;
#pragma unroll
            for (int t = 0; t < 4; t++) w4[t] = o[db][4 * g4 + t] * f;
            *(f32x4 *)(ob + r * Cfg::O_ROWB + (32 * (db + dhalf * NDB) + 8 * g4 + 4 * hh) * 4) = w4;
        }
    if (wave == 0 && dhalf == 0 && hh == 0 && q0 + r < sqv) {
        // softmax.h:179: LSE = max*scale + log(sum); empty -> +inf (final) / -inf (split partial)
        *lse_ptr(q0 + r) = bad ? (is_split ? -INFINITY : INFINITY) : m_all * p.scale + __logf(l_all);
    }
    __syncthreads();
    // each thread sums 8 consecutive d of one row over the waves
    for (int c = tid_h; c < DEC_BM * CPR; c += NT_H) {
        const int row = c / CPR, ch = c % CPR;
        f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < DEC_NW; w++) {
            const char *src = area + (w * DEC_BM + row) * Cfg::O_ROWB + ch * 32;
            a0 += *(const f32x4 *)src;
            a1 += *(const f32x4 *)(src + 16);
        }
        if (q0 + row < sqv && ch * 8 < p.d) {
            if (is_split) {
                float *dst = oacc_ptr(q0 + row) + ch * 8;
                *(f32x4 *)dst = a0;
                *(f32x4 *)(dst + 4) = a1;
            } else {
                i32x4 w;
                w[0] = pack2<BF16>(a0[0], a0[1]); w[1] = pack2<BF16>(a0[2], a0[3]);
                w[2] = pack2<BF16>(a1[0], a1[1]); w[3] = pack2<BF16>(a1[2], a1[3]);
                *(i32x4 *)(o_ptr(q0 + row) + ch * 8) = w;
            }
        }
    }
}

// LSE-weighted merge of the splits: combine_attn_seqk_parallel, flash_fwd_kernel.h:1092-1250.
// Rows = b*h*seqlen_q laid out like oaccum.
template <bool BF16>
__global__ void __launch_bounds__(256)
fwd_combine_kernel(const KParams p) {
    // One workgroup per output row (batch, head, query).  The 256 threads cover (16-byte output
    // chunk) x (split lane): every thread sums the partial outputs of the splits congruent to its
    // lane and the lanes are added through LDS in index order.  Everything the row needs from
    // memory is requested up front -- thread s the LSE of split s, every thread the first PRE
    // partial outputs of its lane -- so the kernel waits for memory once; the maximum and the sum
    // of the LSE terms are wave reductions in a fixed order and the weight of a split is formed
    // once, by its thread.  (Each thread looping over all splits for the maximum, the sum and again
    // for its weights, with the loads behind that: 9.4 us for one sequence of 64 splits.)
    __shared__ float s_w[128];
    __shared__ float s_red[4];
    __shared__ float s_part[256 * 8];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int cpr = p.d / 8;
    const int nlane = 256 / cpr;                 // split lanes
    const int ns = p.num_splits;
    const int64_t total_rows = (int64_t)p.b * p.h * p.seqlen_q;
    const int64_t row = blockIdx.x;
    const int sqi = (int)(row % p.seqlen_q);
    const int bh = (int)(row / p.seqlen_q);
    const int bidb = bh / p.h, bidh = bh % p.h;
    const BlockInfo bi(p, bidb);
    if (sqi >= bi.sq) return;

    const float my = tid < ns ? p.lseaccum[tid * total_rows + row] : -INFINITY;
    const int ch = tid % cpr, sl = tid / cpr;
    constexpr int PRE = 4;
    f32x4 r0[PRE], r1[PRE];
#pragma unroll
    for (int i = 0; i < PRE; i++) {
        const int sp = min(sl + i * nlane, ns - 1);   // (clamped: a valid address, weight 0 below)
        const float *src = p.oaccum + (sp * total_rows + row) * p.d + ch * 8;
        r0[i] = *(const f32x4 *)src;
        r1[i] = *(const f32x4 *)(src + 4);
    }

    auto wave_max = [](float x) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
        return x;
    };
    auto wave_sum = [](float x) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        return x;
    };
    const float m_w = wave_max(my);
    if (lane == 0) s_red[wave] = m_w;
    __syncthreads();
    float lmax = fmaxf(s_red[0], s_red[1]);                  // splits <= 128: waves 0 and 1
    if (lmax == -INFINITY) lmax = 0.f;                       // flash_fwd_kernel.h:1156
    __syncthreads();
    const float e_w = wave_sum(tid < ns ? expf(my - lmax) : 0.f);
    if (lane == 0) s_red[wave] = e_w;
    __syncthreads();
    const float lsum = s_red[0] + s_red[1];
    const float lse = (lsum == 0.f || lsum != lsum) ? INFINITY : logf(lsum) + lmax;  // :1164
    if (tid == 0) p.lse[row] = lse;
    if (tid < 128) s_w[tid] = tid < ns ? expf(my - lse) : 0.f;   // :1172
    __syncthreads();

    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    if (sl < nlane) {
#pragma unroll
        for (int i = 0; i < PRE; i++) {
            const int sp = sl + i * nlane;
            const float w = sp < ns ? s_w[sp < 128 ? sp : 0] : 0.f;
            if (sp < ns) {
                a0 += r0[i] * w;
                a1 += r1[i] * w;
            }
        }
        for (int sp = sl + PRE * nlane; sp < ns; sp += nlane) {
            const float w = s_w[sp];
            const float *src = p.oaccum + (sp * total_rows + row) * p.d + ch * 8;
            a0 += *(const f32x4 *)src * w;
            a1 += *(const f32x4 *)(src + 4) * w;
        }
    }
    *(f32x4 *)(s_part + tid * 8) = a0;
    *(f32x4 *)(s_part + tid * 8 + 4) = a1;
    __syncthreads();
    if (tid < cpr) {
        // lanes are added in index order: the result does not depend on timing
        for (int l = 1; l < nlane; l++) {
            a0 += *(const f32x4 *)(s_part + (l * cpr + tid) * 8);
            a1 += *(const f32x4 *)(s_part + (l * cpr + tid) * 8 + 4);
        }
        unsigned short *og = (unsigned short *)p.o + bi.q_off(p.o_bs, p.o_rs, bidb) +
                             (int64_t)bidh * p.o_hs + (int64_t)sqi * p.o_rs + tid * 8;
        i32x4 w;
        w[0] = pack2<BF16>(a0[0], a0[1]); w[1] = pack2<BF16>(a0[2], a0[3]);
        w[2] = pack2<BF16>(a1[0], a1[1]); w[3] = pack2<BF16>(a1[2], a1[3]);
        *(i32x4 *)og = w;
    }
}

// Append knew/vnew to the cache at [seqlen_k_cache, seqlen_k_cache + seqlen_knew), K rotated with
// its position (flash_fwd_kernel.h:672-782, rotary.h:19-148).  One thread per 8-wide chunk.
template <bool BF16>
__global__ void __launch_bounds__(256)
append_kv_kernel(const KParams p) {
    const int cpr = p.d / 8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)p.b * p.seqlen_knew * p.h_k * cpr;
    if (idx >= total) return;
    const int ch = (int)(idx % cpr);
    int64_t t = idx / cpr;
    const int hk = (int)(t % p.h_k); t /= p.h_k;
    const int tok = (int)(t % p.seqlen_knew);
    const int bidb = (int)(t / p.seqlen_knew);
    const BlockInfo bi(p, bidb);
    const int row = bi.sk_cache + tok;
    if (row >= bi.sk) return;
    const int bidb_cache = p.cache_batch_idx ? p.cache_batch_idx[bidb] : bidb;
    int64_t ko, vo;
    if (p.block_table) {
        const int page = p.block_table[(int64_t)bidb * p.bt_bs + row / p.page_size];
        const int inpage = row % p.page_size;
        ko = (int64_t)page * p.k_bs + (int64_t)inpage * p.k_rs + (int64_t)hk * p.k_hs;
        vo = (int64_t)page * p.v_bs + (int64_t)inpage * p.v_rs + (int64_t)hk * p.v_hs;
    } else {
        ko = (int64_t)bidb_cache * p.k_bs + (int64_t)row * p.k_rs + (int64_t)hk * p.k_hs;
        vo = (int64_t)bidb_cache * p.v_bs + (int64_t)row * p.v_rs + (int64_t)hk * p.v_hs;
    }
    const unsigned short *ks = (const unsigned short *)p.knew + (int64_t)bidb * p.knew_bs +
                               (int64_t)tok * p.knew_rs + (int64_t)hk * p.knew_hs;
    const unsigned short *vs = (const unsigned short *)p.vnew + (int64_t)bidb * p.vnew_bs +
                               (int64_t)tok * p.vnew_rs + (int64_t)hk * p.vnew_hs;
    unsigned short *kd = (unsigned short *)p.k + ko, *vd = (unsigned short *)p.v + vo;
    *(i32x4 *)(vd + ch * 8) = *(const i32x4 *)(vs + ch * 8);
    const int d0 = ch * 8;
    const int half = p.rotary_dim / 2;
    const unsigned short *cosp = (const unsigned short *)p.rotary_cos + (int64_t)row * half;
    const unsigned short *sinp = (const unsigned short *)p.rotary_sin + (int64_t)row * half;
    *(i32x4 *)(kd + d0) = rotary_chunk<BF16>(ks, d0, cosp, sinp, p.rotary_dim, p.rotary_interleaved);
}

// The same merge for few splits (<= 16): one thread per 16-byte output chunk, looping over the
// splits itself -- with so few terms the loop is short and its loads are independent, and the grid
// is rows * d / 8 threads instead of one 256-thread workgroup (two barriers, mostly idle lanes) per
// row.  Sums run in split order, like the kernel above with a single split lane.
template <bool BF16, int N>
__device__ __forceinline__ void combine_flat_row(const KParams &p, const int64_t total_rows, const int64_t row,
                                                 const int ch, unsigned short *og) {
    // every load of the thread is requested before the first result is used (indices clamped to
    // the last split: valid addresses, the terms past it get weight 0), so it waits for memory once
    float ls[N];
    f32x4 r0[N], r1[N];
    const int ns = p.num_splits;
#pragma unroll
    for (int s = 0; s < N; s++) {
        const int sp = min(s, ns - 1);
        ls[s] = p.lseaccum[sp * total_rows + row];
        const float *src = p.oaccum + (sp * total_rows + row) * p.d + ch * 8;
        r0[s] = *(const f32x4 *)src;
        r1[s] = *(const f32x4 *)(src + 4);
    }
    float lmax = -INFINITY;
#pragma unroll
    for (int s = 0; s < N; s++) {
        if (s >= ns) ls[s] = -INFINITY;
        lmax = fmaxf(lmax, ls[s]);
    }
    if (lmax == -INFINITY) lmax = 0.f;
    float lsum = 0.f;
#pragma unroll
    for (int s = 0; s < N; s++)
        if (s < ns) lsum += expf(ls[s] - lmax);
    const float lse = (lsum == 0.f || lsum != lsum) ? INFINITY : logf(lsum) + lmax;
    if (ch == 0) p.lse[row] = lse;
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < N; s++) {
        if (s < ns) {
            const float w = expf(ls[s] - lse);
            a0 += r0[s] * w;
            a1 += r1[s] * w;
        }
    }
    i32x4 w;
    w[0] = pack2<BF16>(a0[0], a0[1]); w[1] = pack2<BF16>(a0[2], a0[3]);
    w[2] = pack2<BF16>(a1[0], a1[1]); w[3] = pack2<BF16>(a1[2], a1[3]);
    *(i32x4 *)og = w;
}

template <bool BF16>
__global__ void __launch_bounds__(256)
fwd_combine_flat_kernel(const KParams p) {
    const int cpr = p.d / 8;
    const int64_t total_rows = (int64_t)p.b * p.h * p.seqlen_q;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total_rows * cpr) return;
    const int64_t row = idx / cpr;
    const int ch = (int)(idx % cpr);
    const int sqi = (int)(row % p.seqlen_q);
    const int bh = (int)(row / p.seqlen_q);
    const int bidb = bh / p.h, bidh = bh % p.h;
    const BlockInfo bi(p, bidb);
    if (sqi >= bi.sq) return;
    unsigned short *og = (unsigned short *)p.o + bi.q_off(p.o_bs, p.o_rs, bidb) +
                         (int64_t)bidh * p.o_hs + (int64_t)sqi * p.o_rs + ch * 8;
    // (sums run in split order whatever the bound)
    if (p.num_splits <= 4) combine_flat_row<BF16, 4>(p, total_rows, row, ch, og);
    else if (p.num_splits <= 8) combine_flat_row<BF16, 8>(p, total_rows, row, ch, og);
    else combine_flat_row<BF16, 16>(p, total_rows, row, ch, og);
}

static hipError_t launch_combine(const KParams &p, hipStream_t stream) {
    const long long rows = (long long)p.b * p.h * p.seqlen_q;
    if (rows == 0) return hipSuccess;
    if (p.num_splits <= 16) {
        const unsigned blocks = (unsigned)((rows * (p.d / 8) + 255) / 256);
        if (p.is_bf16) hipLaunchKernelGGL(fwd_combine_flat_kernel<true>, dim3(blocks), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(fwd_combine_flat_kernel<false>, dim3(blocks), dim3(256), 0, stream, p);
    } else {
        if (p.is_bf16) hipLaunchKernelGGL(fwd_combine_kernel<true>, dim3((unsigned)rows), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(fwd_combine_kernel<false>, dim3((unsigned)rows), dim3(256), 0, stream, p);
    }
    return hipGetLastError();
}

// Heads per workgroup (512 contiguous bytes per key) when the query and KV heads pair up one to one.
template <int DP>
static int decode_heads_per_block(const KParams &p) {
    constexpr int want = DP == 128 ? 2 : 1;
    if (want == 1 || g_tuning[1] == 1) return 1;
    return ((p.h == p.h_k || p.pack_gqa) && p.h_k % want == 0) ? want : 1;
}

template <int DP, bool BF16, int HPB>
static hipError_t launch_decode_hpb(const KParams &p, hipStream_t stream) {
    using Cfg = DecodeCfg<DP, HPB>;
    const int rows = p.seqlen_q * (p.pack_gqa ? p.h_ratio : 1);
    const int n_mblocks = (rows + DEC_BM - 1) / DEC_BM;
    const long long blocks = (long long)n_mblocks * p.b * ((p.pack_gqa ? p.h_k : p.h) / HPB) * p.num_splits;
    if (blocks <= 0) return hipSuccess;
    auto kern = fwd_decode_kernel<DP, BF16, HPB>;
    static std::atomic<uint64_t> lds_set{0};
    if (hipError_t e = set_dyn_lds_once((const void *)kern, Cfg::LDS_BYTES, lds_set); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p, n_mblocks);
    return hipGetLastError();
}

template <int DP, bool BF16>
static hipError_t launch_decode(const KParams &p, hipStream_t stream) {
    hipError_t e;
    constexpr int HPB = DP == 128 ? 2 : 1;
    if (decode_heads_per_block<DP>(p) == 1) e = launch_decode_hpb<DP, BF16, 1>(p, stream);
    else e = launch_decode_hpb<DP, BF16, HPB>(p, stream);
    if (e != hipSuccess) return e;
    if (p.num_splits > 1) e = launch_combine(p, stream);
    return e;
}

// The merge alone, for split partial results written by another kernel (the prefill kernel's
// split variants).
hipError_t run_fwd_combine(const KParams &p, hipStream_t stream) { return launch_combine(p, stream); }

hipError_t run_fwd_splitkv(const KParams &p_in, hipStream_t stream) {
    KParams p = p_in;
    // query heads that share a KV head go into the rows of one workgroup (see the kernel)
    p.pack_gqa = p.h_ratio > 1 && !(g_tuning[2] & 2);
    const bool bf = p.is_bf16;
    if (p.d <= 64) return bf ? launch_decode<64, true>(p, stream) : launch_decode<64, false>(p, stream);
    if (p.d <= 128) return bf ? launch_decode<128, true>(p, stream) : launch_decode<128, false>(p, stream);
    return bf ? launch_decode<256, true>(p, stream) : launch_decode<256, false>(p, stream);
}

hipError_t run_append_kv(const KParams &p, hipStream_t stream) {
    const long long items = (long long)p.b * p.seqlen_knew * p.h_k * (p.d / 8);
    if (items <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((items + 255) / 256);
    if (p.is_bf16) hipLaunchKernelGGL(append_kv_kernel<true>, dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(append_kv_kernel<false>, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace vfa
