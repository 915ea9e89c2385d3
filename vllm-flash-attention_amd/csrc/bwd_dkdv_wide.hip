// dK / dV kernel for head dims above 128 (DP = 256): 4 waves at one wave per SIMD, 32 keys per wave.
//
// Same math and data flow as bwd_dkdv_kernel (bwd.hip; boundary: the reference's Python call sites
// vllm_flash_attn/flash_attn_interface.py:135-154 and csrc/flash_attn/src/flash.h:145-183).  The
// dK^T / dV^T accumulators of 32 keys x 256 columns are 256 registers: the whole accumulation half
// of a wave's 512-entry register file.  Left to the register allocator, accumulator tiles wander
// between the two halves of the file and scratch all through the step (which is why bwd_dkdv_kernel
// cuts the head dim in two column halves taken by separate workgroups, and pays for it with S and
// dP computed in both: 12 products instead of 8 per key block and query slice).  Here the 16
// accumulator tiles are pinned to the accumulation registers by the MFMA statements themselves
// (vfa_one_wave_helpers.h), everything the vector unit touches lives in the other half, and one
// workgroup holds all 256 columns: 8 products, no duplicate.
//
// With a single wave per SIMD nothing hides an LDS read issued right in front of its MFMA, so every
// operand is requested several MFMAs ahead, and the vector work of a 32-row piece is placed beside
// MFMAs it does not depend on:
//   A  S = Q K^T        (16 MFMAs) | the next slice's direct-to-LDS loads
//   C  dP = dO V^T      (16 MFMAs) | the exponentials of S, one per MFMA
//   E  dV += dO^T P, dK += Q^T dS (32 MFMAs) | P / dS packed to 16 bit
//
// This translation unit is compiled with -mllvm -amdgpu-mfma-vgpr-form=1 (results of the S and dP
// MFMAs in the vector half of the file, where the exponentials read them).
#include "vfa_kernel_params.h"
#include "vfa_one_wave_helpers.h"
#include <type_traits>

namespace vfa {

struct BwdWideCfg {
    static constexpr int DP = 256, NW = 4, NT = NW * 64, CPR = DP / 8;
    static constexpr int KB = NW * 32;                        // keys per workgroup
    static constexpr int KV_BYTES = KB * DP * 2;              // one of K / V
    static constexpr int QM = 32;                             // query rows per step
    static constexpr int QT_BYTES = QM * DP * 2;              // one of Q / dO slice
    static constexpr int QBUF_BYTES = 2 * QT_BYTES + 8 * QM;  // Q, dO, -lse/scale[QM], -delta[QM]
    // LDS map: [K tile | V tile] while the fragments are loaded; then [Q bufs (over K) ... | V tile]
    static constexpr int QSPACE = 2 * QBUF_BYTES > KV_BYTES ? 2 * QBUF_BYTES : KV_BYTES;
    static constexpr int OFF_VT = QSPACE;
    static constexpr int MAIN_LDS = QSPACE + KV_BYTES;
    static constexpr int OUT_ROWB = DP * 2 + 16;
    static constexpr int OUT_LDS = NW * 32 * OUT_ROWB;
    static constexpr int LDS_BYTES = MAIN_LDS > OUT_LDS ? MAIN_LDS : OUT_LDS;
    static constexpr int KV_CH = KB * CPR / NT;               // chunks per thread per K (or V) tile
    static constexpr int RPI = 64 / CPR;                      // rows per wave instruction of a slice load
    static constexpr int NI = QM / (RPI * NW);                // instructions per wave, slice and tensor
    static_assert(QM % (RPI * NW) == 0 && (RPI * NW) == 8, "direct slice loads: rounds of 8 rows");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per workgroup");
};

__device__ __forceinline__ void operand_fence2(i32x4 &a, i32x4 &b) {
    asm volatile("s_nop 1" : "+v"(a), "+v"(b));
}

template <bool BF16, int DV = BwdWideCfg::DP>
__global__ void __launch_bounds__(256, 1)
bwd_dkdv_wide_kernel(const KParams p, const int n_kblocks) {
    using Cfg = BwdWideCfg;
    constexpr int DP = Cfg::DP, CPR = Cfg::CPR, NT = Cfg::NT, KB = Cfg::KB, NW = Cfg::NW;
    // DV <= DP: the head dims this instantiation serves, rounded up to 32.  Tiles, slices and fragments
    // keep the layout of DP columns; the products stop at column DV (the rest multiplies zero padding)
    static_assert(DV % 32 == 0 && DV <= DP && DV > 128, "DV");
    constexpr int KS = DV / 16, NDB = DV / 32, QM = Cfg::QM;
    constexpr int ROWB = DP * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;

    // block -> (batch, kv head, key block): whole (batch, kv head) pairs per XCD (blocks b and b + 8
    // share one under round-robin dispatch), pairs in groups of G per XCD, key blocks heaviest
    // first across a group's pairs -- the order of bwd_dkdv_kernel.
    int n_block, hk, bidb;
    {
        const int P = p.b * p.h_k;
        const int L = blockIdx.x;
        const int full = (P / 8) * 8 * n_kblocks;
        int pair;
        if (L < full) {
            const int x = L & 7, sl = L >> 3;
            const int ppx = P / 8;
            const int G = max(1, min(ppx, (64 + n_kblocks - 1) / n_kblocks));
            const int g = sl / (G * n_kblocks);
            const int within = sl - g * G * n_kblocks;
            const int Gl = min(G, ppx - g * G);
            n_block = within / Gl;
            pair = (g * G + within % Gl) * 8 + x;
        } else {
            const int Lr = L - full, R = P - (P / 8) * 8;
            pair = (P / 8) * 8 + Lr % R;
            n_block = Lr / R;
        }
        hk = pair % p.h_k;
        bidb = pair / p.h_k;
    }

    const BlockInfo bi(p, bidb);
    const int sq = bi.sq, sk = bi.sk;
    const int k0 = n_block * KB;
    if (k0 >= sk) return;
    const int kw0 = k0 + 32 * wave;  // first key of this wave

    // query-slice range that can see this key block
    int m_min = 0, m_max = (sq + QM - 1) / QM;
    if (p.is_causal || p.is_local) m_min = max(0, (k0 + sq - sk - p.wr) / QM);
    if (p.is_local) m_max = min(m_max, (k0 + KB + sq - sk + p.wl + QM - 1) / QM);

    const unsigned short *kg = (const unsigned short *)p.k + bi.k_off(p.k_bs, p.k_rs, bidb) + (int64_t)hk * p.k_hs;
    const unsigned short *vg = (const unsigned short *)p.v + bi.k_off(p.v_bs, p.v_rs, bidb) + (int64_t)hk * p.v_hs;

    // ---- stage K and V into LDS; zero rows past sk and columns past d ----
#pragma unroll
    for (int half = 0; half < 2; half++) {
        i32x4 kk[Cfg::KV_CH / 2], vv[Cfg::KV_CH / 2];
#pragma unroll
        for (int i = 0; i < Cfg::KV_CH / 2; i++) {
            const int c = tid + (half * (Cfg::KV_CH / 2) + i) * NT;
            const int row = c / CPR, ch = c % CPR;
            const bool ok = (k0 + row) < sk && ch * 8 < p.d;
            kk[i] = ok ? *(const i32x4 *)(kg + (int64_t)(k0 + row) * p.k_rs + ch * 8) : i32x4{0, 0, 0, 0};
            vv[i] = ok ? *(const i32x4 *)(vg + (int64_t)(k0 + row) * p.v_rs + ch * 8) : i32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < Cfg::KV_CH / 2; i++) {
            const int c = tid + (half * (Cfg::KV_CH / 2) + i) * NT;
            const int off = lds_off<DP>(c / CPR, c % CPR);
            *(i32x4 *)(smem + off) = kk[i];
            *(i32x4 *)(smem + Cfg::OFF_VT + off) = vv[i];
        }
    }
    __syncthreads();
    // B-operand fragments of this wave's keys: lane (r, hh) holds K[kw0 + r][16 s + 8 hh ..]
    i32x4 kf[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) kf[s] = *(const i32x4 *)(smem + lds_off<DP>(32 * wave + r, 2 * s + hh));
    const char *vt = smem + Cfg::OFF_VT + (32 * wave) * ROWB;   // this wave's V rows (fragments are re-read per step)

    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int i = 0; i < NDB; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) { dk[i][j] = 0.f; dv[i][j] = 0.f; }

    const float c2 = p.scale_log2;
    const bool has_alibi = p.alibi_slopes != nullptr;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
    const int n_steps_per_head = max(m_max - m_min, 0);
    const int nh = p.h_ratio;
    const int total_steps = n_steps_per_head * nh;

    // Step -> (32-row slice, query head of the GQA group): slices from the last one down, all heads
    // of the group on a slice before the next slice (the co-resident key blocks of a pair then read
    // the same (slice, head) at about the same time and share it through L2).
    auto advance = [&](int &h_, int &m_) {
        if (++h_ == nh) {
            h_ = 0;
            --m_;
        }
    };
    // Q / dO slices go from global memory straight into LDS (1 KiB = two rows per wave instruction,
    // in lane order, each lane fetching the chunk the XOR swizzle places at its slot); the
    // descriptor ends at the last valid query row, so the row tail and the head-dim padding arrive
    // as zeros.  The rows of a wave's instructions are 8 apart: the lane's swizzle alternates
    // between two values, one for the even and one for the odd instructions.
    unsigned dq_voff[2], ddo_voff[2];
    {
        const int row0 = Cfg::RPI * wave + lane / CPR;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int ch = (lane % CPR) ^ swz<DP>(row0 + 8 * e);
            const bool ok = ch * 8 < p.d;
            dq_voff[e] = ok ? (unsigned)((row0 * (int)p.q_rs + ch * 8) * 2) : 0x80000000u;
            ddo_voff[e] = ok ? (unsigned)((row0 * (int)p.do_rs + ch * 8) * 2) : 0x80000000u;
        }
    }
    auto load_slice = [&](const int gh, const int m, const int buf) __attribute__((always_inline)) {
        const int bidh = hk * p.h_ratio + gh, q0 = m * QM;
        const unsigned short *qg = (const unsigned short *)p.q + bi.q_off(p.q_bs, p.q_rs, bidb) + (int64_t)bidh * p.q_hs;
        const unsigned short *dog = (const unsigned short *)p.dout + bi.q_off(p.do_bs, p.do_rs, bidb) + (int64_t)bidh * p.do_hs;
        const i32x4 q_rsrc = make_rsrc(qg, (unsigned)(((int64_t)(sq - 1) * p.q_rs + p.d) * 2));
        const i32x4 do_rsrc = make_rsrc(dog, (unsigned)(((int64_t)(sq - 1) * p.do_rs + p.d) * 2));
        const unsigned qb = (unsigned)(buf * Cfg::QBUF_BYTES) + (unsigned)wave * 1024u;  // dynamic LDS starts at 0
#pragma unroll
        for (int i = 0; i < Cfg::NI; i++) {
            const unsigned rows = (unsigned)(q0 + 8 * i);
            dma_16b_to_lds(q_rsrc, dq_voff[i & 1] + rows * (unsigned)p.q_rs * 2, qb + NW * i * 1024);
            dma_16b_to_lds(do_rsrc, ddo_voff[i & 1] + rows * (unsigned)p.do_rs * 2, qb + Cfg::QT_BYTES + NW * i * 1024);
        }
    };
    // Row constants of the slice: threads 0..31 carry -LSE / scale (the S accumulator starts from
    // it, so P = exp2(S * c)), threads 32..63 carry -delta (the dP accumulator starts from it).
    float rowc = 0.f;
    auto load_rowc = [&](const int gh, const int m) {
        const int bidh = hk * p.h_ratio + gh, q0 = m * QM;
        if (tid < 2 * QM) {
            const int row = q0 + (tid & (QM - 1));
            const bool ok = row < sq;
            if (tid < QM) rowc = ok ? p.lse[((int64_t)bidb * p.h + bidh) * p.seqlen_q + row] : INFINITY;
            else rowc = ok ? p.dsoftmax_sum[((int64_t)bidb * p.h + bidh) * p.seqlen_q_rounded + row] : 0.f;
        }
    };
    auto store_rowc = [&](int buf) {
        char *qb = smem + buf * Cfg::QBUF_BYTES;
        if (tid < 2 * QM) ((float *)(qb + 2 * Cfg::QT_BYTES))[tid] = tid < QM ? -rowc / p.scale : -rowc;
    };

    int cur_h = 0, cur_m = m_max - 1, nxt_h = 0, nxt_m = m_max - 1;
    advance(nxt_h, nxt_m);
    if (total_steps > 0) load_rowc(cur_h, cur_m);
    __syncthreads();  // every wave has its K fragments: the K tile's space becomes the Q buffers
    if (total_steps > 0) {
        load_slice(cur_h, cur_m, 0);
        store_rowc(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Per-lane LDS offsets of the operand reads (loop invariant).  The swizzle touches the low four
    // bits of the chunk index only, so k-step ks + 8 and d block db + 4 are 256 bytes further on.
    int row_rd[8];           // row read of k-step ks: row r, chunk 2 ks + hh (Q / dO slices, V rows)
#pragma unroll
    for (int ks = 0; ks < 8; ks++) row_rd[ks] = lds_off<DP>(r, 2 * ks + hh);
    int tr_rd[4][2];         // transposed read of d block db: rows 4 hh + tr_q (+ 8), + 16 per k-step
#pragma unroll
    for (int db = 0; db < 4; db++)
#pragma unroll
        for (int e = 0; e < 2; e++)
            tr_rd[db][e] = lds_off<DP>(4 * hh + tr_q + 8 * e, 4 * db + 2 * tr_g + (tr_p >> 1)) + 8 * (tr_p & 1);
    auto rd = [&](const int ks) __attribute__((always_inline)) { return row_rd[ks & 7] + 256 * (ks >> 3); };

    int buf = 0;
    for (int step = 0; step < total_steps; step++) {
        const int bidh = hk * p.h_ratio + cur_h, q0 = cur_m * QM;
        const bool has_next = step + 1 < total_steps;
        // Loads the compiler counts (row constants, ALiBi slope) are issued -- and the slope waited
        // for -- BEFORE the slice loads it does not count: its waits for them then cannot fall
        // behind those and drain them.
        float alibi = 0.f;
        if (has_alibi) alibi = p.alibi_slopes[(int64_t)bidb * p.alibi_bs + bidh] / p.scale;
        if (has_next) load_rowc(nxt_h, nxt_m);
        __builtin_amdgcn_sched_barrier(0);
        // The next slice's descriptors (zero records when there is none: its loads then fetch
        // nothing and need no branch around them).  The loads are spread over the S MFMAs: issued
        // in one burst they cost the wave ~100 cycles each with nothing beside them.
        i32x4 q_rsrc, do_rsrc;
        unsigned q_row0[2], do_row0[2];
        {
            const int bidh_n = hk * p.h_ratio + nxt_h;
            const unsigned short *qg = (const unsigned short *)p.q + bi.q_off(p.q_bs, p.q_rs, bidb) + (int64_t)bidh_n * p.q_hs;
            const unsigned short *dog = (const unsigned short *)p.dout + bi.q_off(p.do_bs, p.do_rs, bidb) + (int64_t)bidh_n * p.do_hs;
            q_rsrc = make_rsrc(qg, has_next ? (unsigned)(((int64_t)(sq - 1) * p.q_rs + p.d) * 2) : 0u);
            do_rsrc = make_rsrc(dog, has_next ? (unsigned)(((int64_t)(sq - 1) * p.do_rs + p.d) * 2) : 0u);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                q_row0[e] = dq_voff[e] + (unsigned)(nxt_m * QM) * (unsigned)p.q_rs * 2;
                do_row0[e] = ddo_voff[e] + (unsigned)(nxt_m * QM) * (unsigned)p.do_rs * 2;
            }
        }
        const unsigned nb = (unsigned)((buf ^ 1) * Cfg::QBUF_BYTES) + (unsigned)wave * 1024u;  // dynamic LDS starts at 0
        auto dma_round = [&](const int i) __attribute__((always_inline)) {
            const unsigned rows = (unsigned)(8 * i);
            dma_16b_to_lds(q_rsrc, q_row0[i & 1] + rows * (unsigned)p.q_rs * 2, nb + NW * i * 1024);
            dma_16b_to_lds(do_rsrc, do_row0[i & 1] + rows * (unsigned)p.do_rs * 2, nb + Cfg::QT_BYTES + NW * i * 1024);
        };
        const char *qb = smem + buf * Cfg::QBUF_BYTES;
        const char *dob = qb + Cfg::QT_BYTES;
        const float *lse_s = (const float *)(qb + 2 * Cfg::QT_BYTES);
        const float *dl_s = lse_s + QM;

        constexpr int PRE = 4;   // operand fragments requested this many MFMAs ahead
        i32x4 qa[KS], da[KS], vb[KS];
        f32x4 ls[4];
#pragma unroll
        for (int ks = 0; ks < PRE; ks++) qa[ks] = *(const i32x4 *)(qb + rd(ks));
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) ls[g4] = *(const f32x4 *)(lse_s + 8 * g4 + 4 * hh);

        // Pieces none of whose entries is visible are computed like any other: the mask zeroes
        // their P, and the step lasts as long as its busiest wave anyway (the waves of a workgroup
        // meet at the step's barrier; steps no wave can see are cut by m_min / m_max).
        const bool masked = has_alibi || (q0 + 32 > sq) || (kw0 + 32 > sk) ||
                            ((p.is_causal || p.is_local) && (kw0 + 32 > q0 + 1 + sk - sq + p.wr)) ||
                            (p.is_local && (kw0 < q0 + 31 + sk - sq - p.wl));
        f32x16 s, dp;
        i32x4 pf[2], dsf[2];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++)
#pragma unroll
            for (int t = 0; t < 4; t++) s[4 * g4 + t] = ls[g4][t];
        // ---- A: S[q][key] = Q K^T - LSE[q] / scale, key on the lane ----
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if (ks + PRE < KS) qa[ks + PRE] = *(const i32x4 *)(qb + rd(ks + PRE));
            else {
                da[ks + PRE - KS] = *(const i32x4 *)(dob + rd(ks + PRE - KS));
                vb[ks + PRE - KS] = *(const i32x4 *)(vt + rd(ks + PRE - KS));
            }
            s = mma32<BF16>(qa[ks], kf[ks], s);
#pragma unroll
            for (int i = 0; i < Cfg::NI; i++)
                if (ks == 1 + i * (KS - 4) / Cfg::NI) dma_round(i);   // (k-steps 1, 4, 7, 10 of 16)
        }
        // The forward's mask and bias, on pieces that need them (wave-uniform).  Register j of lane
        // (r, hh) is local row roff(j) = (j & 3) + 8 (j >> 2) (a constant) + 4 hh of the slice, the
        // lane's key is `col`.  Visible (mask.h:171-172) iff
        //   col <  min(sk, row + 1 + sk - sq + wr)   <=>  roff >= col - shift - wr   (causal / local)
        //   col >= row + sk - sq - wl                 <=>  roff <= col - shift + wl   (local)
        //   row <  sq                                 <=>  roff <  sq - q0 - 4 hh
        // with shift = q0 + 4 hh + sk - sq: one subtract and one unsigned compare per score against
        // the lane's [lo_r, lo_r + span] window.
        if (masked) {
            constexpr int BIG = 1 << 28;
            const int shift = q0 + 4 * hh + sk - sq;
            const int col = kw0 + r;
            int lo_r = (p.is_causal || p.is_local) ? col - shift - p.wr : -BIG;
            int hi_r = p.is_local ? col - shift + p.wl : BIG;
            hi_r = min(hi_r, sq - q0 - 4 * hh - 1);
            lo_r = max(lo_r, -BIG);
            if (col >= sk || hi_r < lo_r) { lo_r = BIG; hi_r = BIG; }  // nothing visible
            const unsigned span = (unsigned)(hi_r - lo_r);
            if (has_alibi) {
                if (p.is_causal) {  // mask.h:180-181: + slope * col
                    const float ab = alibi * (float)col;
#pragma unroll
                    for (int j = 0; j < 16; j++) s[j] += ab;
                } else {            // mask.h:183: - slope * |row + sk - sq - col|
                    const int dc = shift - col;
#pragma unroll
                    for (int j = 0; j < 16; j++) s[j] -= alibi * fabsf((float)((j & 3) + 8 * (j >> 2) + dc));
                }
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int roff = (j & 3) + 8 * (j >> 2);
                s[j] = ((unsigned)(roff - lo_r) <= span) ? s[j] : -INFINITY;
            }
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const f32x4 dl = *(const f32x4 *)(dl_s + 8 * g4 + 4 * hh);
#pragma unroll
            for (int t = 0; t < 4; t++) dp[4 * g4 + t] = dl[t];
        }
        // transposed fragments of the slice for the dV / dK products: group g = (ks2, db)
        i32x4 dot[3], qt[3];
        auto load_tr = [&](const int g, const int slot) __attribute__((always_inline)) {
            const int ks2 = g / NDB, db = g % NDB;
            const int o0 = tr_rd[db & 3][0] + 256 * (db >> 2) + 16 * ks2 * ROWB;
            const int o1 = tr_rd[db & 3][1] + 256 * (db >> 2) + 16 * ks2 * ROWB;
            const i32x2 a0 = __builtin_bit_cast(i32x2, lds_read_tr16(dob + o0));
            const i32x2 a1 = __builtin_bit_cast(i32x2, lds_read_tr16(dob + o1));
            const i32x2 b0 = __builtin_bit_cast(i32x2, lds_read_tr16(qb + o0));
            const i32x2 b1 = __builtin_bit_cast(i32x2, lds_read_tr16(qb + o1));
            dot[slot] = i32x4{a0[0], a0[1], a1[0], a1[1]};
            qt[slot] = i32x4{b0[0], b0[1], b1[0], b1[1]};
        };
        // 16-bit B operands: k-step ks2 (16 query rows) = registers 8 ks2 .. 8 ks2 + 7
        auto pack = [&](const int ks2) __attribute__((always_inline)) {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int j = 8 * ks2 + 2 * w;
                pf[ks2][w] = pack2<BF16>(s[j], s[j + 1]);
                dsf[ks2][w] = pack2<BF16>(s[j] * dp[j], s[j + 1] * dp[j + 1]);
            }
        };
        // ---- C: dP'[q][key] = dO V^T - delta[q]; the exponentials of S beside the MFMAs ----
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if (ks + PRE < KS) {
                da[ks + PRE] = *(const i32x4 *)(dob + rd(ks + PRE));
                vb[ks + PRE] = *(const i32x4 *)(vt + rd(ks + PRE));
            }
            dp = mma32<BF16>(da[ks], vb[ks], dp);
#pragma unroll
            for (int j = ks * 16 / KS; j < (ks + 1) * 16 / KS; j++) s[j] = __builtin_amdgcn_exp2f(s[j] * c2);   // (one per MFMA at KS = 16)
            if (ks == KS - 2) load_tr(0, 0);
            if (ks == KS - 1) load_tr(1, 1);
        }
        asm volatile("" : "+v"(s));
        pack(0);
        operand_fence2(pf[0], dsf[0]);
        // ---- E: dV^T += dO^T P,  dK^T += Q^T dS  (A operands by transposed reads of the slice) ----
        constexpr int NG = 2 * NDB;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int ks2 = g / NDB, db = g % NDB;
            if (g == NDB) operand_fence2(pf[1], dsf[1]);
            if (g + 2 < NG) load_tr(g + 2, (g + 2) % 3);
            mma32_agpr<BF16>(dv[db], dot[g % 3], pf[ks2]);
            mma32_agpr<BF16>(dk[db], qt[g % 3], dsf[ks2]);
            if (g == 0) pack(1);
        }

        if (has_next) store_rowc(buf ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the next slice is in LDS
        __syncthreads();
        buf ^= 1;
        cur_h = nxt_h;
        cur_m = nxt_m;
        advance(nxt_h, nxt_m);
    }

    // ---- write dK (scaled) and dV of this wave's 32 keys, through LDS as whole rows ----
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMA results have landed before they are read
    __syncthreads();
    char *ob = smem + wave * 32 * Cfg::OUT_ROWB;
    unsigned short *dkg = (unsigned short *)p.dk + bi.k_off(p.dk_bs, p.dk_rs, bidb) + (int64_t)hk * p.dk_hs;
    unsigned short *dvg = (unsigned short *)p.dv + bi.k_off(p.dv_bs, p.dv_rs, bidb) + (int64_t)hk * p.dv_hs;
#pragma unroll
    for (int which = 0; which < 2; which++) {
#pragma unroll
        for (int db = 0; db < NDB; db++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                i32x2 w2;
                if (which == 0) {
                    w2[0] = pack2<BF16>(dk[db][4 * g4] * p.scale, dk[db][4 * g4 + 1] * p.scale);
                    w2[1] = pack2<BF16>(dk[db][4 * g4 + 2] * p.scale, dk[db][4 * g4 + 3] * p.scale);
                } else {
                    w2[0] = pack2<BF16>(dv[db][4 * g4], dv[db][4 * g4 + 1]);
                    w2[1] = pack2<BF16>(dv[db][4 * g4 + 2], dv[db][4 * g4 + 3]);
                }
                *(i32x2 *)(ob + r * Cfg::OUT_ROWB + (32 * db + 8 * g4 + 4 * hh) * 2) = w2;
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's own LDS writes
        __builtin_amdgcn_wave_barrier();
        unsigned short *dst = which == 0 ? dkg : dvg;
        const int64_t rs = which == 0 ? p.dk_rs : p.dv_rs;
#pragma unroll
        for (int i = 0; i < (32 * CPR) / 64; i++) {
            const int c = lane + 64 * i;
            const int row = c / CPR, ch = c % CPR;
            const i32x4 w = *(const i32x4 *)(ob + row * Cfg::OUT_ROWB + ch * 16);
            if (kw0 + row < sk && ch * 8 < p.d) *(i32x4 *)(dst + (int64_t)(kw0 + row) * rs + ch * 8) = w;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // the reads above are done before the staging rows are rewritten
        __builtin_amdgcn_wave_barrier();
    }
}

template <bool BF16, int DV>
static hipError_t launch_wide(const KParams &p, hipStream_t stream) {
    using Cfg = BwdWideCfg;
    const int n_kblocks = (p.seqlen_k + Cfg::KB - 1) / Cfg::KB;
    const long long blocks = (long long)n_kblocks * p.b * p.h_k;
    if (blocks <= 0) return hipSuccess;
    auto kern = bwd_dkdv_wide_kernel<BF16, DV>;
    static std::atomic<uint64_t> lds_set{0};
    if (hipError_t e = set_dyn_lds_once((const void *)kern, Cfg::LDS_BYTES, lds_set); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p, n_kblocks);
    return hipGetLastError();
}

// 128 < d <= 256, no dropout, no head split; the caller has checked that the Q / dO extents fit a
// 32-bit buffer offset.
hipError_t run_bwd_dkdv_wide(const KParams &p, hipStream_t stream) {
    // head dims up to 160 / 192 stop their products at that column
    if (p.d <= 160) return p.is_bf16 ? launch_wide<true, 160>(p, stream) : launch_wide<false, 160>(p, stream);
    if (p.d <= 192) return p.is_bf16 ? launch_wide<true, 192>(p, stream) : launch_wide<false, 192>(p, stream);
    return p.is_bf16 ? launch_wide<true, 256>(p, stream) : launch_wide<false, 256>(p, stream);
}

}  // namespace vfa
