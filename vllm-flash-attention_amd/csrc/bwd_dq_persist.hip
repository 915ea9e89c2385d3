// FlashAttention-2 backward, dQ kernel with PERSISTENT workgroups (gfx950 / MI355X).
//
// The query-stationary kernel of bwd.hip (bwd_dq_kernel: a wave owns 32 query rows, Q^T and dO^T
// fragments in registers, K/V tiles of 64 keys stream through LDS, S^T = K Q^T and
// dP^T = V dO^T - delta with the query on the lane, dQ^T += K^T dS^T) with one 8-wave workgroup per
// CU that walks a list of (batch, head, 256-row block) items -- the reference ships no backward
// kernel (setup.py:154-169); the boundary is vllm_flash_attn/flash_attn_interface.py:135-154.
//
// What a block costs outside its tile loop is what the walk removes: with one block per workgroup
// and one workgroup per CU, 6.3 us of prologue (kernel arguments, block mapping, the memory round
// trip of the Q / dO rows and of the first tile) and 1.3 us of epilogue stand against a 91 us loop
// at seqlen 4096.  Here, during the last tile of a block every wave requests its next 32 Q and dO
// rows (16 KiB per wave into LDS images: waves 0 and 1 use the idle tile buffers, the others the
// space behind the tiles -- 160 KiB in all, so the next block's first tile cannot travel at the
// same time) and its next LSE / delta values; at the seam the fragments are read out, the first
// tile is requested into the buffer that has just been freed, and dQ is staged through the
// wave's image space and stored while that tile is in flight.
//
// Work distribution: static, as in fwd_prefill_persist.hip (positions of the launch order with
// L % 8 == x go to the workgroups with blockIdx % 8 == x; workgroup c takes position c of every
// even round of 32 and position 31 - c of every odd one).
//
// Scope: dense batches, 64 < d <= 128, causal or full attention, no ALiBi / window / dropout.
#include "vfa_kernel_params.h"
#include <type_traits>

namespace vfa {

struct DqPersistCfg {
    static constexpr int DP = 128, NW = 8, NT = NW * 64, BM = NW * 32, BN = 64, CPR = DP / 8;
    static constexpr int TILE_BYTES = BN * DP * 2;
    static constexpr int KV_LDS = 4 * TILE_BYTES;               // K0 K1 V0 V1
    static constexpr int IMG_BYTES = 2 * 32 * DP * 2;           // a wave's 32 Q rows, then its 32 dO rows
    static constexpr int O_ROWB = DP * 2 + 16;                  // padded dQ staging row (in the image space)
    static constexpr int LDS_BYTES = KV_LDS + (NW - 2) * IMG_BYTES;
    static_assert(32 * O_ROWB <= IMG_BYTES, "dQ staging fits the image");
    static_assert(IMG_BYTES == TILE_BYTES, "waves 0 and 1 keep their images in the idle tile buffers");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per workgroup");
};

template <bool BF16, int DV = DqPersistCfg::DP>
__global__ void __launch_bounds__(DqPersistCfg::NT, 1)
bwd_dq_persist_kernel(const KParams p, const int n_mblocks, const int n_items) {
    using Cfg = DqPersistCfg;
    constexpr int DP = Cfg::DP, NW = Cfg::NW, NT = Cfg::NT, BM = Cfg::BM, BN = Cfg::BN, CPR = Cfg::CPR;
    // DV <= DP: head dims up to DV; the tiles keep the layout of DP columns, the products stop at DV
    static_assert(DV % 32 == 0 && DV <= DP && 2 * DV > DP, "DV");
    constexpr int KS = DV / 16, NDB = DV / 32;
    constexpr int RPI = 64 / CPR, NI = BN / (RPI * NW);
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The later-dispatched wave of each SIMD loses the issue arbitration (by age) in every phase;
    // a static priority for that half, set once, measured 0.4-1.1 % less time.
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int r = lane & 31, hh = lane >> 5;
    const int r_ = r, hh_ = hh, lane_ = lane, tid_ = tid;   // (shadowed by opaque copies in the seam code)
    const int sq = p.seqlen_q, sk = p.seqlen_k;   // dense batch
    const float c2 = p.scale_log2;

    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwx = gridDim.x >> 3;
    const int n_x = (n_items - xcd + 7) >> 3;
    const int rounds = (n_x + nwx - 1) / nwx;

    struct Blk { int bidb, bidh, m_block, n_max; };
    // This workgroup's next block with visible keys; a block that sees none gets dQ = 0 here.
    auto next_item = [&](int &t, Blk &b) __attribute__((always_inline)) -> bool {
        for (;;) {
            if (++t >= rounds) return false;
            const int s = t * nwx + ((t & 1) ? nwx - 1 - wg : wg);
            if (s >= n_x) continue;
            map_block_at(p, n_mblocks, 32, 8 * s + xcd, b.bidb, b.bidh, b.m_block);
            b.n_max = (sk + BN - 1) / BN;
            if (p.is_causal) b.n_max = min(b.n_max, ((b.m_block + 1) * BM + sk - sq + BN - 1) / BN);
            if (b.n_max > 0) return true;
            unsigned short *dqg = (unsigned short *)p.dq + (int64_t)b.bidb * p.dq_bs + (int64_t)b.bidh * p.dq_hs;
            int tid = tid_;   // (opaque: see the epilogue)
            asm volatile("" : "+v"(tid));
            for (int c = tid; c < BM * CPR; c += NT) {
                const int row = b.m_block * BM + c / CPR, ch = c % CPR;
                if (row < sq && ch * 8 < p.d) *(i32x4 *)(dqg + (int64_t)row * p.dq_rs + ch * 8) = i32x4{0, 0, 0, 0};
            }
        }
    };

    // ---- loop-invariant per-lane offsets (see fwd_prefill_persist.hip) ----
    const unsigned k_rs2 = (unsigned)p.k_rs * 2, v_rs2 = (unsigned)p.v_rs * 2;
    const unsigned q_rs2 = (unsigned)p.q_rs * 2, do_rs2 = (unsigned)p.do_rs * 2;
    unsigned dk_voff, dv_voff;
    {
        const int row0 = RPI * wave + lane / CPR;
        const int ch = (lane % CPR) ^ swz<DP>(row0);
        const bool ok = ch * 8 < p.d;
        dk_voff = ok ? (unsigned)(row0 * (int)k_rs2 + ch * 16) : 0x80000000u;
        dv_voff = ok ? (unsigned)(row0 * (int)v_rs2 + ch * 16) : 0x80000000u;
    }
    const unsigned k_bytes = (unsigned)(((int64_t)(sk - 1) * p.k_rs + p.d) * 2);
    const unsigned v_bytes = (unsigned)(((int64_t)(sk - 1) * p.v_rs + p.d) * 2);
    const unsigned q_bytes = (unsigned)(((int64_t)(sq - 1) * p.q_rs + p.d) * 2);
    const unsigned do_bytes = (unsigned)(((int64_t)(sq - 1) * p.do_rs + p.d) * 2);
    auto kv_rsrc = [&](const Blk &b, const void *base, int64_t bs, int64_t hs, unsigned bytes)
                       __attribute__((always_inline)) {
        const unsigned short *g = (const unsigned short *)base + (int64_t)b.bidb * bs + (int64_t)(b.bidh / p.h_ratio) * hs;
        return __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, bytes, 0x00020000);
    };
    auto load_tile = [&](const __amdgpu_buffer_rsrc_t kr, const __amdgpu_buffer_rsrc_t vr, const int n, auto BUFC)
                         __attribute__((always_inline)) {
        constexpr int BUF = decltype(BUFC)::value;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const unsigned rows = (unsigned)(n * BN + RPI * NW * i);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(kr, (lds_ptr)(smem + BUF * Cfg::TILE_BYTES + (wave + NW * i) * 1024), 16,
                                                     dk_voff + rows * k_rs2, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vr, (lds_ptr)(smem + (2 + BUF) * Cfg::TILE_BYTES + (wave + NW * i) * 1024), 16,
                                                     dv_voff + rows * v_rs2, 0, 0, 0);
        }
    };
    // A wave's image while tile buffer pair OTHER is idle: wave 0 in its K buffer, wave 1 in its V
    // buffer, waves 2-7 behind the tile buffers.
    auto image_of = [&](auto OTHERC) __attribute__((always_inline)) -> char * {
        constexpr int OTHER = decltype(OTHERC)::value;
        return smem + (wave == 0 ? OTHER * Cfg::TILE_BYTES
                       : wave == 1 ? (2 + OTHER) * Cfg::TILE_BYTES
                                   : Cfg::KV_LDS + (wave - 2) * Cfg::IMG_BYTES);
    };
    // lane -> (row, chunk) of its slot in a 1-KiB request; chunks in the head-dim padding are
    // redirected to the last real one (they meet zero columns of K and V, finite stays finite)
    const unsigned col0 = (unsigned)(((lane % CPR) ^ swz<DP>(lane / CPR)) << 4);
    const unsigned col_max = (unsigned)(p.d / 8 - 1) * 16u;
    const unsigned lrow_q = (unsigned)(lane / CPR) * q_rs2, lrow_do = (unsigned)(lane / CPR) * do_rs2;
    auto load_images = [&](const Blk &b, char *fb) __attribute__((always_inline)) {
        const unsigned short *qg = (const unsigned short *)p.q + (int64_t)b.bidb * p.q_bs + (int64_t)b.bidh * p.q_hs;
        const unsigned short *dog = (const unsigned short *)p.dout + (int64_t)b.bidb * p.do_bs + (int64_t)b.bidh * p.do_hs;
        const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)qg, 0, q_bytes, 0x00020000);
        const auto do_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dog, 0, do_bytes, 0x00020000);
        const unsigned row0 = (unsigned)(b.m_block * BM + wave * 32);
        // (opaque: the sixteen per-lane offsets below are loop invariants the compiler would
        //  otherwise keep in registers through the tile loop, which then spills)
        unsigned c0o = col0, lq = lrow_q, ld = lrow_do;
        asm volatile("" : "+v"(c0o), "+v"(lq), "+v"(ld));
#pragma unroll
        for (int i = 0; i < 32 / RPI; i++) {
            const unsigned col = min(c0o ^ (unsigned)(swz<DP>(RPI * i) << 4), col_max);
            const unsigned rws = row0 + RPI * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rsrc, (lds_ptr)(fb + i * 1024), 16, col + lq + rws * q_rs2, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(do_rsrc, (lds_ptr)(fb + 32 * DP * 2 + i * 1024), 16,
                                                     col + ld + rws * do_rs2, 0, 0, 0);
        }
    };
    // LSE and delta of this lane's query row (+inf / 0 past the last row: P = 0 there)
    auto load_rowc = [&](const Blk &b, float &lse_v, float &dl_v) __attribute__((always_inline)) {
        int r = r_;   // (opaque: see the epilogue)
        asm volatile("" : "+v"(r));
        const int row = b.m_block * BM + wave * 32 + r;
        const bool rok = row < sq;
        lse_v = rok ? p.lse[((int64_t)b.bidb * p.h + b.bidh) * p.seqlen_q + row] : INFINITY;
        dl_v = rok ? p.dsoftmax_sum[((int64_t)b.bidb * p.h + b.bidh) * p.seqlen_q_rounded + row] : 0.f;
    };
    // LDS read offsets, built once (rows 16 or 32 further down keep the row's swizzle: those steps
    // are immediates): k_rd = row read of k-step ks (row r, chunk 2 ks + hh), used for the K and V
    // tiles and for the Q / dO images; t_rd = transposed read of d block db (rows 4 hh + tr_q, + 8)
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
    constexpr int ROWB = DP * 2;
    int k_rd[KS], t_rd[NDB][2];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) k_rd[ks] = lds_off<DP>(r, 2 * ks + hh);
#pragma unroll
    for (int db = 0; db < NDB; db++)
#pragma unroll
        for (int e = 0; e < 2; e++)
            t_rd[db][e] = lds_off<DP>(4 * hh + tr_q + 8 * e, 4 * db + 2 * tr_g + (tr_p >> 1)) + 8 * (tr_p & 1);

    // ---- first block ----
    Blk cur, nxt;
    int t = -1;
    if (!next_item(t, cur)) return;
    auto k_rsrc = kv_rsrc(cur, p.k, p.k_bs, p.k_hs, k_bytes);
    auto v_rsrc = kv_rsrc(cur, p.v, p.v_bs, p.v_hs, v_bytes);
    load_tile(k_rsrc, v_rsrc, cur.n_max - 1, std::integral_constant<int, 0>{});
    load_images(cur, image_of(std::integral_constant<int, 1>{}));
    float lse_n, dl_n;   // row constants of the block whose images are in flight
    load_rowc(cur, lse_n, dl_n);
    bool have_next = next_item(t, nxt);

    i32x4 qf[KS], dof[KS];
    // the dP chain's start, -delta, as a block of 16 registers kept for the whole row block, or
    // rebuilt per 32 keys: here the registers are worth more (with the block the compiler has one
    // set of operand registers left and every MFMA waits for its own LDS read)
#ifndef VFA_DQP_NDL
#define VFA_DQP_NDL 0
#endif
    constexpr bool NDL_BLOCK = VFA_DQP_NDL;
    f32x16 dq[NDB], ndl;
    float lse2, dl_r;
    int n, q0w;
    // fragments out of the wave's image (its loads have landed: vmcnt(0) precedes every call)
    auto read_frags = [&](const char *fb) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < KS; s++) {
            qf[s] = *(const i32x4 *)(fb + k_rd[s]);
            dof[s] = *(const i32x4 *)(fb + 32 * ROWB + k_rd[s]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read out before the image space is written again
    };
    // accumulators and row constants of block `cur` (whose LSE / delta are in lse_n, dl_n)
    auto reset_block = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NDB; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) dq[i][j] = 0.f;
        lse2 = lse_n * (float)M_LOG2E;
        dl_r = dl_n;
        if constexpr (NDL_BLOCK) {
#pragma unroll
            for (int j = 0; j < 16; j++) ndl[j] = -dl_r;
            asm volatile("" : "+v"(ndl));   // kept as 16 registers: the dP chain starts from this block
        }
        n = cur.n_max - 1;
        q0w = cur.m_block * BM + wave * 32;
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    read_frags(image_of(std::integral_constant<int, 1>{}));
    reset_block();
    __syncthreads();

    // dq = round16(scale * dQ) of block `b`, staged through `ob` (the wave's image space) as whole rows
    auto epilogue = [&](const Blk &b, char *ob) __attribute__((always_inline)) {
        unsigned short *dqg = (unsigned short *)p.dq + (int64_t)b.bidb * p.dq_bs + (int64_t)b.bidh * p.dq_hs;
        // (opaque copies of the lane's coordinates: every per-lane address below is an invariant of
        //  the whole walk, and what the compiler hoists out of it stays in registers through the
        //  tile loop, whose schedule degrades with every register it loses)
        int r = r_, hh = hh_, lane = lane_;
        asm volatile("" : "+v"(r), "+v"(hh), "+v"(lane));
#pragma unroll
        for (int db = 0; db < NDB; db++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                i32x2 w;
                w[0] = pack2<BF16>(dq[db][4 * g4 + 0] * p.scale_rp, dq[db][4 * g4 + 1] * p.scale_rp);
                w[1] = pack2<BF16>(dq[db][4 * g4 + 2] * p.scale_rp, dq[db][4 * g4 + 3] * p.scale_rp);
                *(i32x2 *)(ob + r * Cfg::O_ROWB + (32 * db + 8 * g4 + 4 * hh) * 2) = w;
            }
        // (the wave's own staging rows, LDS operations of a wave in order: no barrier)
#pragma unroll
        for (int i = 0; i < (32 * CPR) / 64; i++) {
            const int c = lane + 64 * i;
            const int row = c / CPR, ch = c % CPR;
            const i32x4 w = *(const i32x4 *)(ob + row * Cfg::O_ROWB + ch * 16);
            if (q0w + row < sq && ch * 8 < p.d) *(i32x4 *)(dqg + (int64_t)(q0w + row) * p.dq_rs + ch * 8) = w;
        }
    };

    enum { CONT = 0, SEAM = 1, DONE = 2 };
    // One K/V tile; BUF (compile time) is the LDS buffer pair it sits in.  CONT: the next tile is in
    // the other pair; SEAM: a new block begins, its first tile is in THIS pair; DONE: no block left.
    auto tile_step = [&](auto BUFC) __attribute__((always_inline)) -> int {
        constexpr int BUF = decltype(BUFC)::value;
        using Other = std::integral_constant<int, BUF ^ 1>;
        const bool last = n == 0;
        if (!last) {
            load_tile(k_rsrc, v_rsrc, n - 1, Other{});
        } else if (have_next) {
            load_images(nxt, image_of(Other{}));
            load_rowc(nxt, lse_n, dl_n);
        }
        const char *kb = smem + BUF * Cfg::TILE_BYTES;
        const char *vb = smem + (2 + BUF) * Cfg::TILE_BYTES;
        const int c0 = n * BN;
        bool need = c0 + BN > sk;
        if (p.is_causal) need = need || (c0 + BN > q0w + 1 + sk - sq);
#pragma unroll
        for (int kb2 = 0; kb2 < 2; kb2++) {
            // 32-key halves above this wave's diagonal are skipped; the wave still takes part in the staging
            const int ch0 = c0 + 32 * kb2;
            if (p.is_causal && ch0 >= q0w + 32 + sk - sq) continue;
            f32x16 s, dp;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                s[j] = 0.f;
                if constexpr (!NDL_BLOCK) dp[j] = -dl_r;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
#ifdef EXP_INLINE_OFF
                const int off = lds_off<DP>(32 * kb2 + r, 2 * ks + hh);
#else
                const int off = k_rd[ks] + 32 * kb2 * ROWB;
#endif
                s = mma32<BF16>(*(const i32x4 *)(kb + off), qf[ks], s);
                dp = mma32<BF16>(*(const i32x4 *)(vb + off), dof[ks], (NDL_BLOCK && ks == 0) ? ndl : dp);
            }
            if (need) {
                const int row = q0w + r;
                const int hi = p.is_causal ? min(sk, row + 1 + sk - sq) : sk;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int col = c0 + 32 * kb2 + (j & 3) + 8 * (j >> 2) + 4 * hh;
                    s[j] = col < hi ? s[j] : -INFINITY;
                }
            }
            i32x4 dsf[2];
#pragma unroll
            for (int j = 0; j < 16; j++) dp[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j], c2, -lse2)) * dp[j];
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
                for (int w = 0; w < 4; w++) dsf[ks2][w] = pack2<BF16>(dp[8 * ks2 + 2 * w], dp[8 * ks2 + 2 * w + 1]);
            // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
            for (int db = 0; db < NDB; db++) {
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ks2++) {
                    const i32x2 a0 = __builtin_bit_cast(i32x2, lds_read_tr16(kb + t_rd[db][0] + (32 * kb2 + 16 * ks2) * ROWB));
                    const i32x2 a1 = __builtin_bit_cast(i32x2, lds_read_tr16(kb + t_rd[db][1] + (32 * kb2 + 16 * ks2) * ROWB));
                    const i32x4 kt = {a0[0], a0[1], a1[0], a1[1]};
                    dq[db] = mma32<BF16>(kt, dsf[ks2], dq[db]);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the next tile, or the next block's images and row constants
        __syncthreads();
        if (!last) {
            --n;
            return CONT;
        }
        // ---- the seam ----
        if (!have_next) {
            epilogue(cur, image_of(Other{}));
            return DONE;
        }
        // the next block's first tile into the pair that has just been freed (every wave has passed
        // the barrier), its fragments out of the image, then this block's dQ through the image space
        k_rsrc = kv_rsrc(nxt, p.k, p.k_bs, p.k_hs, k_bytes);
        v_rsrc = kv_rsrc(nxt, p.v, p.v_bs, p.v_hs, v_bytes);
        load_tile(k_rsrc, v_rsrc, nxt.n_max - 1, BUFC);
        read_frags(image_of(Other{}));
        epilogue(cur, image_of(Other{}));
        cur = nxt;
        reset_block();
        have_next = next_item(t, nxt);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the first tile (and the dQ stores)
        __syncthreads();                       // ... and every wave is done with its image space
        return SEAM;
    };

    for (;;) {
        int rc;
        do { rc = tile_step(std::integral_constant<int, 0>{}); } while (rc == SEAM);
        if (rc == DONE) break;
        do { rc = tile_step(std::integral_constant<int, 1>{}); } while (rc == SEAM);
        if (rc == DONE) break;
    }
}

bool bwd_dq_persist_ok(const KParams &p) {
    if (p.d <= 64 || p.d > 128) return false;
    if (p.cu_seqlens_q || p.cu_seqlens_k || p.seqused_k || p.block_table || p.cache_batch_idx) return false;
    if (p.alibi_slopes || p.is_local) return false;
    if (p.seqlen_q <= 0 || p.seqlen_k <= 0) return false;
    const int64_t lim = (int64_t)1 << 30;
    const int64_t rq = (int64_t)p.seqlen_q + 256, rk = (int64_t)p.seqlen_k + 256;
    return rq * p.q_rs * 2 < lim && rq * p.do_rs * 2 < lim && rk * p.k_rs * 2 < lim && rk * p.v_rs * 2 < lim;
}

long long bwd_dq_persist_items(const KParams &p) {
    return (long long)((p.seqlen_q + DqPersistCfg::BM - 1) / DqPersistCfg::BM) * p.b * p.h;
}

hipError_t run_bwd_dq_persist(const KParams &p, hipStream_t stream) {
    using Cfg = DqPersistCfg;
    const int n_mblocks = (p.seqlen_q + Cfg::BM - 1) / Cfg::BM;
    const long long items = bwd_dq_persist_items(p);
    if (items <= 0) return hipSuccess;
    if (items > 0x7fffffffLL) return hipErrorInvalidValue;
    const int grid = std::max(8, device_num_cus() / 8 * 8);
    // head dims 65..96: the products cut off at column 96 (b8 s4096 h32 d96 causal 1.413 -> 1.235 ms)
    const bool cut = p.d <= 96;
    static std::atomic<uint64_t> lds_set[4];
    const void *kern = p.is_bf16 ? (cut ? (const void *)bwd_dq_persist_kernel<true, 96> : (const void *)bwd_dq_persist_kernel<true>)
                                 : (cut ? (const void *)bwd_dq_persist_kernel<false, 96> : (const void *)bwd_dq_persist_kernel<false>);
    if (hipError_t e = set_dyn_lds_once(kern, Cfg::LDS_BYTES, lds_set[2 * p.is_bf16 + cut]); e != hipSuccess) return e;
    KParams pk = p;
    int nm = n_mblocks, ni = (int)items;
    void *args[] = {&pk, &nm, &ni};
    if (hipError_t e = hipLaunchKernel(kern, dim3((unsigned)grid), dim3(Cfg::NT), args, Cfg::LDS_BYTES, stream); e != hipSuccess)
        return e;
    return hipGetLastError();
}

}  // namespace vfa
