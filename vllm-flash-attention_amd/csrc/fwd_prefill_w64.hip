// FlashAttention-2 forward for long query blocks on gfx950 / MI355X: one wave per SIMD, 64 query
// rows per wave, software-pipelined over key tiles.
//
// Same mathematics and operand layouts as fwd_prefill.hip (the counterpart of
// csrc/flash_attn/src/flash_fwd_kernel.h:27-470 of the reference); what differs is the schedule:
//   * a workgroup is 4 waves x 64 query rows (BM = 256) and may use all 512 vector registers, so
//     every K fragment and every V^T fragment read from LDS feeds TWO 32-row MFMAs: LDS traffic per
//     MFMA is half that of the 32-rows-per-wave kernel, where it equals the matrix-pipe time.
//   * the tile loop is skewed by one stage per product: iteration i issues S^T(i) = K(i) Q^T,
//     runs the softmax of tile i-1 and accumulates O^T += V^T(i-2) P^T(i-2).  The three are
//     independent, so the exponentials / maxima / conversions of one tile sit in the issue gaps
//     of the MFMAs of its neighbours instead of idling the matrix pipe.
//   * the running-max rescale of O is exact (no threshold): it is applied after the pending
//     P V product of the previous tile has been added, and skipped when no lane's max moved.
//   * K/V tiles arrive through buffer loads whose descriptor ends at the last valid key row, so
//     the key tail and the head-dim padding read as zeros with no predication in the loop.
// Paged KV and head dims above 128 stay on fwd_prefill.hip.
#include "vfa_kernel_params.h"
#include <type_traits>

namespace vfa {

#ifdef VFA_STAMP
// Diagnostic build only: per-phase shader-clock sums of the tile loop (never in the product build).
__device__ unsigned long long g_stamps64[4096 * 8];
#define STAMP(slot)                                                        \
    do {                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                 \
        unsigned long long t_;                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                 \
        acc_[slot] += t_ - last_;                                          \
        last_ = t_;                                                        \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

template <int DP>
struct Prefill64Cfg {
    static constexpr int NT = 256;
    static constexpr int BM = 256;
    static constexpr int BN = 64;
    static constexpr int CPR = DP / 8;                     // 16-byte chunks per row
    static constexpr int RSTEP = NT / CPR;                 // rows between a lane's chunks
    static constexpr int CH = BN / RSTEP;                  // chunks per lane per tile
    static constexpr int TILE_BYTES = BN * DP * 2;
    static constexpr int O_ROWB = DP * 2 + 16;             // padded O staging row
    static constexpr int KV_LDS = 6 * TILE_BYTES;          // K0 K1 K2 V0 V1 V2
    static constexpr int O_LDS = BM * O_ROWB;
    static constexpr int LDS_BYTES = KV_LDS > O_LDS ? KV_LDS : O_LDS;
};

template <int DP, bool BF16>
__global__ void __launch_bounds__(256, 1)
fwd_prefill64_kernel(const KParams p, const int n_mblocks) {
    using Cfg = Prefill64Cfg<DP>;
    constexpr int NT = Cfg::NT, BM = Cfg::BM, BN = Cfg::BN, CPR = Cfg::CPR, CH = Cfg::CH;
    constexpr int RSTEP = Cfg::RSTEP, TILE = Cfg::TILE_BYTES;
    constexpr int KS = DP / 16;   // k-steps of QK^T
    constexpr int NDB = DP / 32;  // 32-wide d blocks of O
    constexpr int ROWB = DP * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;

    // ---- block -> (batch, head, m_block): whole (batch, kv head) pairs per XCD (blocks b and
    // b+8 share one), heaviest m-blocks first ----
    int bidb, bidh, m_block;
    {
        const int g = p.h_ratio;
        const int W = g * n_mblocks;
        const int P = p.b * p.h_k;
        const int L = blockIdx.x;
        const int full = (P / 8) * 8 * W;
        int pair, item;
        if (L < full) {
            const int x = L & 7, s = L >> 3;
            pair = (s / W) * 8 + x;
            item = s % W;
        } else {
            const int Lr = L - full;
            pair = (P / 8) * 8 + Lr / W;
            item = Lr % W;
        }
        bidb = pair / p.h_k;
        const int hk0 = pair % p.h_k;
        bidh = hk0 * g + item % g;
        m_block = n_mblocks - 1 - item / g;
    }
    const int hk = bidh / p.h_ratio;

    const BlockInfo bi(p, bidb);
    const int sq = bi.sq, sk = bi.sk;
    if (m_block * BM >= sq) return;

    // ---- key tile range: flash_fwd_kernel.h:59-67 ----
    const int n_min = !p.is_local ? 0 : max(0, (m_block * BM + sk - sq - p.wl) / BN);
    int n_max = (sk + BN - 1) / BN;
    if (p.is_causal || p.is_local)
        n_max = min(n_max, ((m_block + 1) * BM + sk - sq + p.wr + BN - 1) / BN);

    const unsigned short *qg = (const unsigned short *)p.q + bi.q_off(p.q_bs, p.q_rs, bidb) +
                               (int64_t)bidh * p.q_hs;
    unsigned short *og = (unsigned short *)p.o + bi.q_off(p.o_bs, p.o_rs, bidb) +
                         (int64_t)bidh * p.o_hs;
    float *lse_g = p.lse + ((int64_t)bidb * p.h + bidh) * p.seqlen_q;
    const int q0w = m_block * BM + wave * 64;  // first query row of this wave

    if (n_max <= n_min) {
        // No visible key for this row block: O = 0, LSE = +inf (flash_fwd_kernel.h:70-106).
        for (int c = tid; c < BM * CPR; c += NT) {
            const int row = m_block * BM + c / CPR, ch = c % CPR;
            if (row < sq && ch * 8 < p.d)
                *(i32x4 *)(og + (int64_t)row * p.o_rs + ch * 8) = i32x4{0, 0, 0, 0};
        }
        for (int c = tid; c < BM; c += NT)
            if (m_block * BM + c < sq) lse_g[m_block * BM + c] = INFINITY;
        return;
    }
    const int T = n_max - n_min;  // tiles of this block; tile t is keys 64 (n_max-1-t) ..
    // First tile (in processing order) that holds a key visible to any of this wave's rows:
    // tiles above the wave's own diagonal are only staged, not computed.
    int t_w = 0;
    if (p.is_causal || p.is_local) {
        const int lim = q0w + 64 + sk - sq + p.wr;  // first column no row of the wave sees
        const int n_w = lim <= 0 ? 0 : (lim + BN - 1) / BN;
        t_w = n_max - max(n_min, min(n_max, n_w));
    }

    // ---- K/V addressing: buffer descriptors ending at the last valid row ----
    const int bidb_cache = p.cache_batch_idx ? p.cache_batch_idx[bidb] : bidb;
    const unsigned short *kg = (const unsigned short *)p.k + (int64_t)hk * p.k_hs +
                               bi.k_off(p.k_bs, p.k_rs, bidb_cache);
    const unsigned short *vg = (const unsigned short *)p.v + (int64_t)hk * p.v_hs +
                               bi.k_off(p.v_bs, p.v_rs, bidb_cache);
    const unsigned k_bytes = (unsigned)(((int64_t)(sk - 1) * p.k_rs + p.d) * 2);
    const unsigned v_bytes = (unsigned)(((int64_t)(sk - 1) * p.v_rs + p.d) * 2);
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)kg, 0, k_bytes, 0x00020000);
    const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)vg, 0, v_bytes, 0x00020000);
    const int st_row = tid / CPR, st_ch = tid % CPR;
    const bool colok = st_ch * 8 < p.d;
    // out-of-range offset for the padding columns: the load returns zeros
    const unsigned k_voff = colok ? (unsigned)((st_row * (int)p.k_rs + st_ch * 8) * 2) : 0x80000000u;
    const unsigned v_voff = colok ? (unsigned)((st_row * (int)p.v_rs + st_ch * 8) * 2) : 0x80000000u;
    const unsigned k_rs2 = (unsigned)p.k_rs * 2, v_rs2 = (unsigned)p.v_rs * 2;

    i32x4 kreg[CH], vreg[CH];
    auto load_k1 = [&](int n, int i) __attribute__((always_inline)) {
        kreg[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
            k_rsrc, k_voff + (unsigned)(n * BN + RSTEP * i) * k_rs2, 0, 0));
    };
    // the V window that goes with K tile n: rows 64 n + 32 .. 64 n + 95
    auto load_v1 = [&](int n, int i) __attribute__((always_inline)) {
        vreg[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
            v_rsrc, v_voff + (unsigned)(n * BN + 32 + RSTEP * i) * v_rs2, 0, 0));
    };
    auto load_k = [&](int n) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; i++) load_k1(n, i);
    };
    auto load_v = [&](int n) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; i++) load_v1(n, i);
    };
    // quarter G of the loads of a tile: issued one quarter per sub-step, so that the four waves
    // of the workgroup do not queue up behind each other at the address unit after the barrier
    auto load_quarter = [&](int n, auto GC) __attribute__((always_inline)) {
        constexpr int G = decltype(GC)::value;
        if constexpr (CH == 4) {
            load_k1(n, G);
            load_v1(n, G);
        } else {  // CH == 2
            if constexpr (G < 2) load_k1(n, G);
            else load_v1(n, G - 2);
        }
    };
    // a lane's chunks sit RSTEP rows apart, which leaves the XOR swizzle unchanged.  K ring
    // buffers at 0, TILE, 2 TILE; V ring buffers behind them (their base is folded into the
    // per-lane offset so that every access fits the 16-bit immediate of the DS instructions).
    const int st_off = lds_off<DP>(st_row, st_ch);
    const int st_off_v = st_off + 3 * TILE;
    auto store_k = [&](auto BUFC) __attribute__((always_inline)) {
        constexpr int BUF = decltype(BUFC)::value;
#pragma unroll
        for (int i = 0; i < CH; i++)
            *(i32x4 *)(smem + st_off + i * RSTEP * ROWB + BUF * TILE) = kreg[i];
    };
    auto store_v = [&](auto BUFC) __attribute__((always_inline)) {
        constexpr int BUF = decltype(BUFC)::value;
#pragma unroll
        for (int i = 0; i < CH; i++)
            *(i32x4 *)(smem + st_off_v + i * RSTEP * ROWB + BUF * TILE) = vreg[i];
    };

    // ---- Q fragments (B operand of S^T = K Q^T): lane (r, hh) holds Q[q0w + 32 rb + r][16 s + 8 hh ..] ----
    i32x4 qf[2][KS];
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        const int row = q0w + 32 * rb + r;
        const bool rok = row < sq;
        const unsigned short *qr = qg + (int64_t)(rok ? row : 0) * p.q_rs;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int col = 16 * s + 8 * hh;
            qf[rb][s] = (rok && col < p.d) ? *(const i32x4 *)(qr + col) : i32x4{0, 0, 0, 0};
        }
    }
    // Q is only ever a matrix operand: park it in the accumulation half of the register file,
    // the arithmetic half is needed for the softmax.
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int s = 0; s < KS; s++) asm volatile("" : "+a"(qf[rb][s]));

    f32x16 o[2][NDB];
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int i = 0; i < NDB; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) o[rb][i][j] = 0.f;
    float m_run[2] = {-INFINITY, -INFINITY};  // reference point of the exponentials (raw score units)
    float l_run[2] = {0.f, 0.f};              // running row sum, PARTIAL per lane half

    float alibi = 0.f;
    const bool has_alibi = p.alibi_slopes != nullptr;
    if (has_alibi) alibi = p.alibi_slopes[(int64_t)bidb * p.alibi_bs + bidh] / p.scale;
    const float c2 = p.scale_log2;

    // per-lane LDS read offsets
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
    int k_rd[KS];
    int v_rd[NDB][2];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) k_rd[ks] = lds_off<DP>(r, 2 * ks + hh);
#pragma unroll
    for (int db = 0; db < NDB; db++)
#pragma unroll
        for (int e = 0; e < 2; e++)
            v_rd[db][e] = lds_off<DP>(4 * hh + tr_q + 8 * e, 4 * db + 2 * tr_g + (tr_p >> 1)) + 8 * (tr_p & 1) +
                          3 * TILE;

    // The pipeline works on HALF tiles of 32 keys, upper half (keys 64 n + 32 ..) first:
    //   S^T(h) = K(h) Q^T   runs beside   the softmax of half h-1, row block 1
    //   O^T += V^T(h-1) P^T(h-1)   runs beside   the softmax of half h, row block 0
    // so P V lags Q K^T by one half.  The V window staged for an iteration is therefore shifted
    // up by 32 keys against the K tile: iteration i (K tile n) multiplies the lower half of tile
    // n+1 and the upper half of tile n, i.e. V rows 64 n + 32 .. 64 n + 95.
    // Carried across iterations: raw scores of the pending half for row block 1, P^T fragments of
    // row block 0 for it, and the rescale factors the pending half asks for.
    f32x16 s1;           // [reg] scores, row block 1, pending half
    i32x4 pf0[2];        // [k-step] P^T, row block 0, pending half
    float alpha[2] = {1.f, 1.f};
    bool moved = false;                    // wave-uniform
    constexpr float RESCALE_LOG2 = 4.f;

    // softmax of one 32-row block of one half tile (softmax.h:135-166), one query row per lane:
    // sc = raw scores (c0 = first key of the half) -> P^T fragments of its two k-steps.
    // Fragment ks holds regs 8 ks .. 8 ks + 7: element j of lane half hh is key
    // 16 ks + 8 (j>>2) + 4 hh + (j&3).
    // masks and ALiBi for one 32-row block of one half tile (mask.h:128-209); c0 = first key
    auto apply_mask = [&](f32x16 &sc, const int rb, const int c0) __attribute__((always_inline)) {
        const int row = q0w + 32 * rb + r;
        const int lo = p.is_local ? max(0, row + sk - sq - p.wl) : 0;
        const int hi = (p.is_causal || p.is_local) ? min(sk, row + 1 + sk - sq + p.wr) : sk;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int col = c0 + (j & 3) + 8 * (j >> 2) + 4 * hh;
            float x = sc[j];
            if (has_alibi)
                x = p.is_causal ? x + alibi * (float)col
                                : x - alibi * fabsf((float)(row + sk - sq - col));
            sc[j] = (col >= lo && col < hi) ? x : -INFINITY;
        }
    };
    // does the half tile starting at key c0 need apply_mask for this wave's 64 rows?
    auto needs_mask = [&](const int c0) __attribute__((always_inline)) {
#ifdef W64_NO_MASK
        return false;
#endif
        bool need = has_alibi || (c0 + 32 > sk) || (c0 < 0);
        if (p.is_causal || p.is_local) need = need || (c0 + 32 > q0w + 1 + sk - sq + p.wr);
        if (p.is_local) need = need || (c0 < q0w + 63 + sk - sq - p.wl);
        return need;
    };
    // softmax of one 32-row block of one half tile (softmax.h:135-166), one query row per lane:
    // sc = (masked) raw scores -> P^T fragments of its two k-steps.  Fragment ks holds regs
    // 8 ks .. 8 ks + 7: element j of lane half hh is key 16 ks + 8 (j>>2) + 4 hh + (j&3).
    auto softmax_rb = [&](f32x16 sc, const int rb, i32x4 (&pout)[2]) __attribute__((always_inline)) {
        float mx = sc[0];
#pragma unroll
        for (int j = 1; j < 16; j++) mx = fmaxf(mx, sc[j]);
        mx = half_swap_max(mx);
        // The reference point of the exponentials follows the running maximum lazily: it moves
        // (for every row of the wave) only when some row's maximum outgrew it by more than
        // 2^RESCALE_LOG2, so P stays below that bound and the O rescale, which is expensive
        // here, becomes rare.  Any reference point gives the same O and LSE up to rounding.
        const bool upd = __any((mx - m_run[rb]) * c2 > RESCALE_LOG2);
        const float m_new = upd ? fmaxf(m_run[rb], mx) : m_run[rb];
        const float m_cur = (m_new == -INFINITY) ? 0.f : m_new;
        moved = moved || upd;
        alpha[rb] = __builtin_amdgcn_exp2f((m_run[rb] - m_cur) * c2);
        m_run[rb] = m_new;
        const float mc = m_cur * c2;
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[j], c2, -mc));
            sc[j] = e;
            psum += e;
        }
        l_run[rb] = l_run[rb] * alpha[rb] + psum;
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int w = 0; w < 4; w++)
                pout[ks][w] = pack2<BF16>(sc[8 * ks + 2 * w], sc[8 * ks + 2 * w + 1]);
        // pin: without a use here the optimizer sinks the exponentials past the next branch, into
        // the block of the P V product that consumes them, away from the MFMAs meant to hide them
        asm volatile("" ::"v"(pout[0]), "v"(pout[1]));
    };
    // O holds every half before the pending one at the scale of the old maxima: bring it to the
    // maxima of the pending half before that half is added (skipped when no lane's max moved).
    auto rescale_o = [&]() __attribute__((always_inline)) {
#ifdef W64_NO_RESCALE
        if (false) {
#else
        if (__builtin_expect(moved, 0)) {
#endif
            // O sits in accumulation registers, which no arithmetic instruction can touch:
            // read, scale, write back to the SAME register, so that both arms of this rare
            // branch leave O where it was.  The leading idle states let matrix instructions still
            // in flight retire before their results are read, the trailing ones separate the
            // last write from the next matrix instruction.
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int db = 0; db < NDB; db++)
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        float x = o[rb][db][j], t;
                        asm volatile("v_accvgpr_read_b32 %1, %0\n\t"
                                     "v_mul_f32 %1, %1, %2\n\t"
                                     "v_accvgpr_write_b32 %0, %1"
                                     : "+a"(x), "=&v"(t)
                                     : "v"(alpha[rb]));
                        o[rb][db][j] = x;
                    }
            asm volatile("s_nop 7" ::: "memory");
        }
        moved = false;
    };

    // prologue: tiles 0 and 1 into ring buffers 0 and 1
    load_k(n_max - 1);
    load_v(n_max - 1);
    store_k(std::integral_constant<int, 0>{});
    store_v(std::integral_constant<int, 0>{});
    load_k(n_max - 2);
    load_v(n_max - 2);
    store_k(std::integral_constant<int, 1>{});
    store_v(std::integral_constant<int, 1>{});
    __syncthreads();

    // A pending half of all -inf scores / zero P is neutral (max, sum and O stay as they are),
    // so the first iteration needs no special case.
#pragma unroll
    for (int j = 0; j < 16; j++) s1[j] = -INFINITY;
    pf0[0] = pf0[1] = i32x4{0, 0, 0, 0};

    // One wave per SIMD has nobody to hide its LDS latency: operand fragments are read one
    // sub-step ahead of the MFMAs that use them.  K and V fragments alternate, so two register
    // sets suffice: while one feeds the matrix pipe the other is in flight.
    i32x4 kf[KS];        // K half tile: one row-read per k-step
    i32x4 vf[NDB * 2];   // V^T half window: one transposed read pair per (d block, k-step)
    // kb / vb = ring buffer base WITHOUT the per-lane part (and, for V, without the 3 TILE base)
    auto ld_kfrag = [&](const int kbuf_off, auto KHC) __attribute__((always_inline)) {
        constexpr int KH = decltype(KHC)::value;  // 1 = upper keys, local rows 32..63
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
            kf[ks] = *(const i32x4 *)(smem + k_rd[ks] + kbuf_off + 32 * KH * ROWB);
    };
    auto ld_vfrag = [&](const int vbuf_off, auto VHC) __attribute__((always_inline)) {
        constexpr int VH = decltype(VHC)::value;  // 1 = local rows 32..63 = lower half of tile n+1
#pragma unroll
        for (int db = 0; db < NDB; db++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const s16x4 v0 = lds_read_tr16(smem + v_rd[db][0] + vbuf_off + (32 * VH + 16 * ks) * ROWB);
                const s16x4 v1 = lds_read_tr16(smem + v_rd[db][1] + vbuf_off + (32 * VH + 16 * ks) * ROWB);
                const i32x2 a0 = __builtin_bit_cast(i32x2, v0), a1 = __builtin_bit_cast(i32x2, v1);
                vf[db * 2 + ks] = i32x4{a0[0], a0[1], a1[0], a1[1]};
            }
    };
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // S^T of the half tile in kf for both row blocks: sn[rb][reg] = score(key = first key of the
    // half + (reg&3) + 8 (reg>>2) + 4 hh, query = q0w + 32 rb + r)
    auto qk_mma = [&](f32x16 (&sn)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
                sn[rb] = mma32<BF16>(kf[ks], qf[rb][ks], ks == 0 ? zero : sn[rb]);
    };
    // O^T += V^T P^T for the half window in vf
    auto pv_mma = [&](const i32x4 (&p0)[2], const i32x4 (&p1)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int db = 0; db < NDB; db++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                o[0][db] = mma32<BF16>(vf[db * 2 + ks], p0[ks], o[0][db]);
                o[1][db] = mma32<BF16>(vf[db * 2 + ks], p1[ks], o[1][db]);
            }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;

    // Iteration i: K tile n = n_max-1-i and the V window above are in ring buffers RING = i % 3;
    // tile n-2 and its window are staged into buffers (RING + 2) % 3, which the previous
    // iteration was the last to read.  Loads past either end of the key range are harmless: the
    // buffer descriptor turns them into zeros, or they fetch rows nobody multiplies.  Four
    // sub-steps of 16 MFMAs, each beside one 32 x 32 softmax and the fragment reads of the next
    // sub-step; the (rare) mask and rescale branches sit between sub-steps so that each is one
    // straight-line block the scheduler can interleave.
#ifdef VFA_STAMP
    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
    const unsigned long long t_begin_ = last_;
#endif
    auto step = [&](const int i, auto RINGC) __attribute__((always_inline)) {
        constexpr int RING = decltype(RINGC)::value;
        constexpr int KB = RING * TILE, KB_NEXT = ((RING + 1) % 3) * TILE;
        const int n_i = n_max - 1 - i;
        STAMP(0);
        const int c_a = n_i * BN + 32, c_b = n_i * BN;
        f32x16 sa[2], sb[2];
        i32x4 p1[2], pa0[2], pa1[2];

        if (needs_mask(c_a + 32)) apply_mask(s1, 1, c_a + 32);
        load_quarter(n_i - 2, std::integral_constant<int, 0>{});
        ld_vfrag(KB, H1{});
        qk_mma(sa);                 // kf: upper half of tile i, read during the previous iteration
        softmax_rb(s1, 1, p1);
        rescale_o();
        if (needs_mask(c_a)) { apply_mask(sa[0], 0, c_a); apply_mask(sa[1], 1, c_a); }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(1);
        load_quarter(n_i - 2, std::integral_constant<int, 1>{});
        ld_kfrag(KB, H0{});
        pv_mma(pf0, p1);
        softmax_rb(sa[0], 0, pa0);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(2);
        load_quarter(n_i - 2, std::integral_constant<int, 2>{});
        ld_vfrag(KB, H0{});
        qk_mma(sb);
        softmax_rb(sa[1], 1, pa1);
        rescale_o();
        if (needs_mask(c_b)) apply_mask(sb[0], 0, c_b);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3);
        load_quarter(n_i - 2, std::integral_constant<int, 3>{});
        ld_kfrag(KB_NEXT, H1{});
        pv_mma(pa0, pa1);
        softmax_rb(sb[0], 0, pf0);
        s1 = sb[1];
        STAMP(4);

        store_k(std::integral_constant<int, (RING + 2) % 3>{});
        store_v(std::integral_constant<int, (RING + 2) % 3>{});
        STAMP(5);
        __syncthreads();
        STAMP(6);
    };

    // Tiles above the wave's own diagonal are fully masked; they cost a wave its share of the
    // block's staging anyway, so they simply run masked.
    ld_kfrag(0, H1{});
    for (int i = 0;;) {
        step(i, std::integral_constant<int, 0>{});
        if (++i >= T) break;
        step(i, std::integral_constant<int, 1>{});
        if (++i >= T) break;
        step(i, std::integral_constant<int, 2>{});
        if (++i >= T) break;
    }
#ifdef VFA_STAMP
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long *dst = g_stamps64 + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 7; i++) dst[i] = acc_[i];
        dst[7] = (unsigned long long)T;
    }
#endif
    // drain: the lower half of the last tile (keys 64 n_min .. 64 n_min + 31) is still pending;
    // its V rows are the upper half of the window that goes with tile T (ring buffer T % 3).
    {
        i32x4 p1[2];
        apply_mask(s1, 1, n_min * BN);
        softmax_rb(s1, 1, p1);
        rescale_o();
        ld_vfrag((T % 3) * TILE, H1{});
        pv_mma(pf0, p1);
    }
    __syncthreads();  // all LDS reads done before the tiles are reused for the O staging

    // ---- epilogue: normalise, LSE (softmax.h:168-185), store O through LDS as whole rows ----
    // (the loop's last barrier fenced all LDS reads of the K/V tiles)
    char *ob = smem + wave * 64 * Cfg::O_ROWB;
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        const float l_tot = half_swap_sum(l_run[rb]);
        const bool bad = (l_tot == 0.f) || (l_tot != l_tot);
        const float inv = bad ? 1.f : 1.f / l_tot;
        const float lse = bad ? INFINITY : m_run[rb] * p.scale + __logf(l_tot);
        const int row = q0w + 32 * rb + r;
        if (hh == 0 && row < sq) lse_g[row] = lse;
#pragma unroll
        for (int db = 0; db < NDB; db++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                i32x2 w;
                w[0] = pack2<BF16>(o[rb][db][4 * g4 + 0] * inv, o[rb][db][4 * g4 + 1] * inv);
                w[1] = pack2<BF16>(o[rb][db][4 * g4 + 2] * inv, o[rb][db][4 * g4 + 3] * inv);
                *(i32x2 *)(ob + (32 * rb + r) * Cfg::O_ROWB + (32 * db + 8 * g4 + 4 * hh) * 2) = w;
            }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's own LDS writes
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < (64 * CPR) / 64; i++) {
        const int c = lane + 64 * i;
        const int row = c / CPR, ch = c % CPR;
        const i32x4 w = *(const i32x4 *)(ob + row * Cfg::O_ROWB + ch * 16);
        if (q0w + row < sq && ch * 8 < p.d)
            *(i32x4 *)(og + (int64_t)(q0w + row) * p.o_rs + ch * 8) = w;
    }
}

template <int DP, bool BF16>
static hipError_t launch_prefill64(const KParams &p, hipStream_t stream) {
    using Cfg = Prefill64Cfg<DP>;
    const int n_mblocks = (p.seqlen_q + Cfg::BM - 1) / Cfg::BM;
    const long long blocks = (long long)n_mblocks * p.b * p.h;
    if (blocks <= 0) return hipSuccess;
    auto kern = fwd_prefill64_kernel<DP, BF16>;
    static bool attr_set = false;
    if (!attr_set && Cfg::LDS_BYTES > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p,
                       n_mblocks);
    return hipGetLastError();
}

// True when the 64-rows-per-wave kernel can take the problem: contiguous (non-paged) K/V whose
// per-(batch, head) extent fits a 32-bit buffer offset, head dim up to 128.
bool fwd_prefill64_supported(const KParams &p) {
    if (p.block_table || p.d > 128) return false;
    const int64_t lim = (int64_t)1 << 30;  // bytes, with room for the tile offsets added per lane
    const int64_t rows = (int64_t)p.seqlen_k + 256;
    return rows * p.k_rs * 2 < lim && rows * p.v_rs * 2 < lim;
}

hipError_t run_fwd_prefill64(const KParams &p, hipStream_t stream) {
    const bool bf = p.is_bf16;
    if (p.d <= 64) return bf ? launch_prefill64<64, true>(p, stream) : launch_prefill64<64, false>(p, stream);
    return bf ? launch_prefill64<128, true>(p, stream) : launch_prefill64<128, false>(p, stream);
}

}  // namespace vfa

#ifdef VFA_STAMP
extern "C" int vfa_debug_read_stamps64(unsigned long long *host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vfa::g_stamps64), n * sizeof(unsigned long long));
}
#endif
