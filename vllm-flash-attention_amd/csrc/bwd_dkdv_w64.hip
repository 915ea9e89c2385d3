// dK / dV kernel with 64 keys per wave, one wave per SIMD (4 waves, 256 keys per workgroup).
//
// Same math and data flow as bwd_dkdv_kernel (bwd.hip; boundary: the reference's Python call sites
// vllm_flash_attn/flash_attn_interface.py:135-154 and csrc/flash_attn/src/flash.h:145-183), other
// decomposition: a wave owns TWO 32-key halves, so every Q / dO fragment it reads from LDS (row
// reads for S and dP, transposed reads for dV and dK) feeds two MFMAs, and with the whole
// 512-entry register file of its SIMD it keeps the dK^T / dV^T accumulators of 64 keys (256
// registers), the K fragments (64) and still has room to fetch LDS operands several MFMAs ahead.
// In the 8-wave kernel every MFMA waits for an LDS read issued one MFMA earlier (no registers left
// at 2 waves per SIMD); here the exponentials of one key half run beside the MFMAs of the other and
// the operand reads are in flight well before their MFMA.
//
// This translation unit is compiled with -mllvm -amdgpu-mfma-vgpr-form=1: by default a kernel that
// may use more than 256 registers gets every MFMA result in the accumulator half of the file, and
// each score then costs a v_accvgpr_read before the vector unit can touch it.
#include "vfa_kernel_params.h"
#include "vfa_one_wave_helpers.h"
#include <type_traits>


namespace vfa {

#ifdef VFA_STAMP
// Diagnostic build only: per-phase shader-clock sums of the step loop (per block and wave).
__device__ unsigned long long g_stamps_w64[512 * 4 * 8];
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
struct Bwd64Cfg {
    static constexpr int NW = 4, NT = NW * 64, KH = 2;
    static constexpr int KB = NW * 32 * KH;                   // keys per workgroup
    static constexpr int CPR = DP / 8;
    static constexpr int KV_BYTES = KB * DP * 2;              // one of K / V
    static constexpr int QM = 64;                             // query rows per step (two sub-blocks of 32)
    static constexpr int QT_BYTES = QM * DP * 2;              // one of Q / dO slice
    static constexpr int QBUF_BYTES = 2 * QT_BYTES + 8 * QM;  // Q, dO, lse[QM], -delta[QM]
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
    static_assert(QM % (RPI * NW) == 0 && (RPI * NW) % 16 == 0, "direct slice loads");
};

template <int DP, bool BF16>
__global__ void __launch_bounds__(256, 1)
bwd_dkdv_w64_kernel(const KParams p, const int n_kblocks) {
    using Cfg = Bwd64Cfg<DP>;
    constexpr int CPR = Cfg::CPR, NT = Cfg::NT, KB = Cfg::KB, NW = Cfg::NW;
    constexpr int KS = DP / 16, NDB = DP / 32, QM = Cfg::QM;
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
    const int kw0 = k0 + 64 * wave;  // first key of this wave

    // query-slice range that can see this key block
    int m_min = 0, m_max = (sq + QM - 1) / QM;
    if (p.is_causal || p.is_local) m_min = max(0, (k0 + sq - sk - p.wr) / QM);
    if (p.is_local) m_max = min(m_max, (k0 + KB + sq - sk + p.wl + QM - 1) / QM);

    const unsigned short *kg = (const unsigned short *)p.k + bi.k_off(p.k_bs, p.k_rs, bidb) + (int64_t)hk * p.k_hs;
    const unsigned short *vg = (const unsigned short *)p.v + bi.k_off(p.v_bs, p.v_rs, bidb) + (int64_t)hk * p.v_hs;

    // ---- stage K and V into LDS; zero rows past sk and columns past d ----
    {
#pragma unroll
        for (int i = 0; i < Cfg::KV_CH; i++) {
            const int c = tid + i * NT;
            const int row = c / CPR, ch = c % CPR;
            const bool ok = (k0 + row) < sk && ch * 8 < p.d;
            const i32x4 kk = ok ? *(const i32x4 *)(kg + (int64_t)(k0 + row) * p.k_rs + ch * 8) : i32x4{0, 0, 0, 0};
            const i32x4 vv = ok ? *(const i32x4 *)(vg + (int64_t)(k0 + row) * p.v_rs + ch * 8) : i32x4{0, 0, 0, 0};
            const int off = lds_off<DP>(row, ch);
            *(i32x4 *)(smem + off) = kk;
            *(i32x4 *)(smem + Cfg::OFF_VT + off) = vv;
        }
    }
    __syncthreads();
    // B-operand fragments of this wave's keys: lane (r, hh) holds K[kw0 + 32 kh + r][16 s + 8 hh ..]
    i32x4 kf[2][KS];
#pragma unroll
    for (int kh = 0; kh < 2; kh++)
#pragma unroll
        for (int s = 0; s < KS; s++)
            kf[kh][s] = *(const i32x4 *)(smem + lds_off<DP>(64 * wave + 32 * kh + r, 2 * s + hh));
    const char *vt = smem + Cfg::OFF_VT + (64 * wave) * ROWB;   // this wave's V rows (fragments are re-read per step)

    f32x16 dk[2][NDB], dv[2][NDB];
#pragma unroll
    for (int kh = 0; kh < 2; kh++)
#pragma unroll
        for (int i = 0; i < NDB; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) { dk[kh][i][j] = 0.f; dv[kh][i][j] = 0.f; }

    const float c2 = p.scale_log2;
    const bool has_alibi = p.alibi_slopes != nullptr;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
    const int n_steps_per_head = max(m_max - m_min, 0);
    const int nh = p.h_ratio;
    const int total_steps = n_steps_per_head * nh;

    // Step -> (64-row slice, query head of the GQA group): slices from the last one down, all heads
    // of the group on a slice before the next slice (the co-resident key blocks of a pair then read
    // the same (slice, head) at about the same time and share it through L2).
    auto advance = [&](int &h_, int &m_) {
        if (++h_ == nh) {
            h_ = 0;
            --m_;
        }
    };
    // Q / dO slices go from global memory straight into LDS (buffer_load_dwordx4 ... lds: 1 KiB per
    // wave instruction in lane order, each lane fetching the chunk the XOR swizzle places at its
    // slot); the descriptor ends at the last valid query row, so the row tail and the head-dim
    // padding arrive as zeros.
    unsigned dq_voff, ddo_voff;
    {
        const int row0 = Cfg::RPI * wave + lane / CPR;
        const int ch = (lane % CPR) ^ swz<DP>(row0);
        const bool ok = ch * 8 < p.d;
        dq_voff = ok ? (unsigned)((row0 * (int)p.q_rs + ch * 8) * 2) : 0x80000000u;
        ddo_voff = ok ? (unsigned)((row0 * (int)p.do_rs + ch * 8) * 2) : 0x80000000u;
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
            const unsigned rows = (unsigned)(q0 + Cfg::RPI * NW * i);
            dma_16b_to_lds(q_rsrc, dq_voff + rows * (unsigned)p.q_rs * 2, qb + NW * i * 1024);
            dma_16b_to_lds(do_rsrc, ddo_voff + rows * (unsigned)p.do_rs * 2, qb + Cfg::QT_BYTES + NW * i * 1024);
        }
    };
    // Row constants of the slice: threads 0..63 carry -LSE / scale (the S accumulator starts from
    // it, so P = exp2(S * c)), threads 64..127 carry -delta (the dP accumulator starts from it).
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

    // per-lane LDS offsets of the operand reads (loop invariant)
    int row_rd[KS];          // row read of k-step ks: row r, chunk 2 ks + hh (Q / dO slices, V rows + 32 kh rows)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) row_rd[ks] = lds_off<DP>(r, 2 * ks + hh);
    int tr_rd[NDB][2];       // transposed read of d block db: rows 4 hh + tr_q (+ 8), + 16 per k-step
#pragma unroll
    for (int db = 0; db < NDB; db++)
#pragma unroll
        for (int e = 0; e < 2; e++)
            tr_rd[db][e] = lds_off<DP>(4 * hh + tr_q + 8 * e, 4 * db + 2 * tr_g + (tr_p >> 1)) + 8 * (tr_p & 1);

#ifdef VFA_STAMP
    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
    int buf = 0;
    for (int step = 0; step < total_steps; step++) {
        const int bidh = hk * p.h_ratio + cur_h, q0s = cur_m * QM;
        STAMP(7);
        const bool has_next = step + 1 < total_steps;
        // Loads the compiler counts (row constants, ALiBi slope) are issued -- and the slope waited
        // for -- BEFORE the slice loads it does not count: its waits for them then cannot fall
        // behind those and drain them.
        float alibi = 0.f;
        if (has_alibi) alibi = p.alibi_slopes[(int64_t)bidb * p.alibi_bs + bidh] / p.scale;
        if (has_next) load_rowc(nxt_h, nxt_m);
        __builtin_amdgcn_sched_barrier(0);
        // The next slice's descriptors (zero records when there is none: its loads then fetch
        // nothing and the pieces below need no branch around them).  The loads themselves are
        // spread over the step, NI / 2 row rounds per sub-block, between the dV / dK MFMAs: issued
        // in one burst at the top they cost the wave ~100 cycles each with nothing beside them.
        i32x4 q_rsrc, do_rsrc;
        unsigned q_row0, do_row0;
        {
            const int bidh_n = hk * p.h_ratio + nxt_h;
            const unsigned short *qg = (const unsigned short *)p.q + bi.q_off(p.q_bs, p.q_rs, bidb) + (int64_t)bidh_n * p.q_hs;
            const unsigned short *dog = (const unsigned short *)p.dout + bi.q_off(p.do_bs, p.do_rs, bidb) + (int64_t)bidh_n * p.do_hs;
            q_rsrc = make_rsrc(qg, has_next ? (unsigned)(((int64_t)(sq - 1) * p.q_rs + p.d) * 2) : 0u);
            do_rsrc = make_rsrc(dog, has_next ? (unsigned)(((int64_t)(sq - 1) * p.do_rs + p.d) * 2) : 0u);
            q_row0 = dq_voff + (unsigned)(nxt_m * QM) * (unsigned)p.q_rs * 2;
            do_row0 = ddo_voff + (unsigned)(nxt_m * QM) * (unsigned)p.do_rs * 2;
        }
        const unsigned nb = (unsigned)((buf ^ 1) * Cfg::QBUF_BYTES) + (unsigned)wave * 1024u;  // dynamic LDS starts at 0
        auto dma_round = [&](const int i) __attribute__((always_inline)) {
            const unsigned rows = (unsigned)(Cfg::RPI * NW * i);
            dma_16b_to_lds(q_rsrc, q_row0 + rows * (unsigned)p.q_rs * 2, nb + NW * i * 1024);
            dma_16b_to_lds(do_rsrc, do_row0 + rows * (unsigned)p.do_rs * 2, nb + Cfg::QT_BYTES + NW * i * 1024);
        };
        const char *qb0 = smem + buf * Cfg::QBUF_BYTES;
        const float *lse_s0 = (const float *)(qb0 + 2 * Cfg::QT_BYTES);

        // One wave per SIMD: nothing hides an LDS read that is issued right before its MFMA, so every
        // operand is requested well ahead of its use: the Q fragments and row constants of a
        // sub-block under the dV / dK MFMAs of the one before it (the first of a step right after
        // the barrier), the first PRE k-steps of dO / V fragments under the S MFMAs, the rest PRE
        // k-steps ahead under the dP MFMAs, the transposed fragments two groups of four MFMAs ahead.
        constexpr int PRE = KS < 3 ? KS : 3;
        auto fetch_s_ops = [&](const int sub, i32x4 (&qa)[KS], f32x4 (&ls)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) qa[ks] = *(const i32x4 *)(qb0 + 32 * sub * ROWB + row_rd[ks]);
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) ls[g4] = *(const f32x4 *)(lse_s0 + 32 * sub + 8 * g4 + 4 * hh);
        };
        // One 32-row x 64-key piece.  SUB: which sub-block; qa / ls: its prefetched S operands;
        // qa_n / ls_n: where to prefetch those of the next piece (NEXT >= 0), DMA0..DMA1: the row
        // rounds of the next slice to request here.
        auto piece = [&](auto SUBC, auto NEXTC, i32x4 (&qa)[KS], f32x4 (&ls)[4], i32x4 (&qa_n)[KS], f32x4 (&ls_n)[4])
                         __attribute__((always_inline)) {
            constexpr int sub = decltype(SUBC)::value, NEXT = decltype(NEXTC)::value;
            constexpr int DMA0 = sub == 1 ? 0 : Cfg::NI / 2, DMA1 = sub == 1 ? Cfg::NI / 2 : Cfg::NI;
            const int q0 = q0s + 32 * sub;
            const char *qb = qb0 + 32 * sub * ROWB;
            const char *dob = qb + Cfg::QT_BYTES;
            const float *dl_s = lse_s0 + QM + 32 * sub;
            // Pieces none of whose entries is visible are computed like any other: the mask zeroes
            // their P, and the step lasts as long as its busiest wave anyway (the waves of a
            // workgroup meet at the step's barrier; steps no wave can see are cut by m_min / m_max).
            bool masked = has_alibi || (q0 + 32 > sq);
#pragma unroll
            for (int kh = 0; kh < 2; kh++) {
                const int kw = kw0 + 32 * kh;
                masked = masked || (kw + 32 > sk) ||
                         ((p.is_causal || p.is_local) && (kw + 32 > q0 + 1 + sk - sq + p.wr)) ||
                         (p.is_local && (kw < q0 + 31 + sk - sq - p.wl));
            }
            f32x16 s[2], dp[2];
            i32x4 pf[2][2], dsf[2][2];
            i32x4 da[KS], vb0[KS], vb1[KS];
            // ---- S[q][key] = Q K^T - LSE[q] / scale, key on the lane, one key half after the other
            // (the Q fragments stay in registers for both) ----
            // the forward's mask and bias for one key half, on pieces that need them (wave-uniform).
            // Register j of lane (r, hh) is local row roff(j) = (j & 3) + 8 (j >> 2) (a constant)
            // + 4 hh of the sub-block, the lane's key is `col`.  Visible (mask.h:171-172) iff
            //   col <  min(sk, row + 1 + sk - sq + wr)   <=>  roff >= col - shift - wr   (causal / local)
            //   col >= row + sk - sq - wl                 <=>  roff <= col - shift + wl   (local)
            //   row <  sq                                 <=>  roff <  sq - q0 - 4 hh
            // with shift = q0 + 4 hh + sk - sq: one subtract and one unsigned compare per score
            // against the lane's [lo_r, lo_r + span] window.
            auto apply_mask = [&](const int kh) __attribute__((always_inline)) {
                constexpr int BIG = 1 << 28;
                const int shift = q0 + 4 * hh + sk - sq;
                const int col = kw0 + 32 * kh + r;
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
                        for (int j = 0; j < 16; j++) s[kh][j] += ab;
                    } else {            // mask.h:183: - slope * |row + sk - sq - col|
                        const int dc = shift - col;
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            s[kh][j] -= alibi * fabsf((float)((j & 3) + 8 * (j >> 2) + dc));
                    }
                }
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int roff = (j & 3) + 8 * (j >> 2);
                    s[kh][j] = ((unsigned)(roff - lo_r) <= span) ? s[kh][j] : -INFINITY;
                }
            };
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++)
#pragma unroll
                for (int t = 0; t < 4; t++) { s[0][4 * g4 + t] = ls[g4][t]; s[1][4 * g4 + t] = ls[g4][t]; }
            // Vector work is placed beside MFMAs it does not depend on, phase by phase, so that no
            // phase asks for more issue slots than its MFMAs leave (about 24 cycles each):
            //   A  S of key half 0          | the next slice's direct-to-LDS loads of this piece
            //   B  S of key half 1          | exponentials of half 0
            //   C  dP of both key halves    | exponentials of half 1
            //   E  dV / dK (k-step 0, 1)    | P / dS packed to 16 bit (k-step 1 under k-step 0's MFMAs)
            // (LLVM would sink each group of exponentials to the block of its first use: the empty
            // asm statements pin them to their phase.)
            // ---- A ----
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                s[0] = mma32<BF16>(qa[ks], kf[0][ks], s[0]);
#pragma unroll
                for (int i = DMA0; i < DMA1; i++)
                    if (ks == 1 + 2 * (i - DMA0)) dma_round(i);
            }
            if (masked) apply_mask(0);
            // ---- B ----
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                s[1] = mma32<BF16>(qa[ks], kf[1][ks], s[1]);
#pragma unroll
                for (int e = ks * 16 / KS; e < (ks + 1) * 16 / KS; e++)
                    s[0][e] = __builtin_amdgcn_exp2f(s[0][e] * c2);
                if (ks >= KS - PRE) {
                    da[ks - (KS - PRE)] = *(const i32x4 *)(dob + row_rd[ks - (KS - PRE)]);
                    vb0[ks - (KS - PRE)] = *(const i32x4 *)(vt + row_rd[ks - (KS - PRE)]);
                }
            }
            asm volatile("" : "+v"(s[0]));
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const f32x4 dl = *(const f32x4 *)(dl_s + 8 * g4 + 4 * hh);
#pragma unroll
                for (int t = 0; t < 4; t++) { dp[0][4 * g4 + t] = dl[t]; dp[1][4 * g4 + t] = dl[t]; }
            }
            STAMP(1);
            if (masked) apply_mask(1);
            STAMP(2);
            // transposed fragments of the slice for the dV / dK products: group g = (ks2, db)
            i32x4 dot[3], qt[3];
            auto load_tr = [&](const int g, const int slot) __attribute__((always_inline)) {
                const int ks2 = g / NDB, db = g % NDB;
                const int o0 = tr_rd[db][0] + 16 * ks2 * ROWB, o1 = tr_rd[db][1] + 16 * ks2 * ROWB;
                const i32x2 a0 = __builtin_bit_cast(i32x2, lds_read_tr16(dob + o0));
                const i32x2 a1 = __builtin_bit_cast(i32x2, lds_read_tr16(dob + o1));
                const i32x2 b0 = __builtin_bit_cast(i32x2, lds_read_tr16(qb + o0));
                const i32x2 b1 = __builtin_bit_cast(i32x2, lds_read_tr16(qb + o1));
                dot[slot] = i32x4{a0[0], a0[1], a1[0], a1[1]};
                qt[slot] = i32x4{b0[0], b0[1], b1[0], b1[1]};
            };
            // 16-bit B operands: k-step ks2 (16 query rows) = registers 8 ks2 .. 8 ks2 + 7
            auto pack = [&](const int kh, const int ks2) __attribute__((always_inline)) {
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const int j = 8 * ks2 + 2 * w;
                    pf[kh][ks2][w] = pack2<BF16>(s[kh][j], s[kh][j + 1]);
                    dsf[kh][ks2][w] = pack2<BF16>(s[kh][j] * dp[kh][j], s[kh][j + 1] * dp[kh][j + 1]);
                }
            };
            // ---- C/D: dP'[q][key] = dO V^T - delta[q], both key halves interleaved (a dO fragment
            // feeds two MFMAs in a row), the exponentials of half 1 beside them ----
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                if (ks < PRE) vb1[ks] = *(const i32x4 *)(vt + 32 * ROWB + row_rd[ks]);
                if (ks + PRE < KS) {
                    da[ks + PRE] = *(const i32x4 *)(dob + row_rd[ks + PRE]);
                    vb0[ks + PRE] = *(const i32x4 *)(vt + row_rd[ks + PRE]);
                    vb1[ks + PRE] = *(const i32x4 *)(vt + 32 * ROWB + row_rd[ks + PRE]);
                }
                dp[0] = mma32<BF16>(da[ks], vb0[ks], dp[0]);
                dp[1] = mma32<BF16>(da[ks], vb1[ks], dp[1]);
#pragma unroll
                for (int e = ks * 16 / KS; e < (ks + 1) * 16 / KS; e++)
                    s[1][e] = __builtin_amdgcn_exp2f(s[1][e] * c2);
                if (ks == KS - 2) load_tr(0, 0);
                if (ks == KS - 1) load_tr(1, 1);
            }
            asm volatile("" : "+v"(s[1]));
            STAMP(3);
            pack(0, 0);
            pack(1, 0);
            operand_fence(pf[0][0], dsf[0][0], pf[1][0], dsf[1][0]);
            pack(0, 1);
            // ---- E: dV^T += dO^T P,  dK^T += Q^T dS  (A operands by transposed reads of the slice,
            // shared by the two key halves) ----
            constexpr int NG = 2 * NDB;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const int ks2 = g / NDB, db = g % NDB;
                if (g == NDB) operand_fence(pf[0][1], dsf[0][1], pf[1][1], dsf[1][1]);
                if (g + 2 < NG) load_tr(g + 2, (g + 2) % 3);
                if (NEXT >= 0 && g == NG - 2) fetch_s_ops(NEXT, qa_n, ls_n);  // all transposed reads are out by now
#pragma unroll
                for (int kh = 0; kh < 2; kh++) {
                    mma32_agpr<BF16>(dv[kh][db], dot[g % 3], pf[kh][ks2]);
                    mma32_agpr<BF16>(dk[kh][db], qt[g % 3], dsf[kh][ks2]);
                }
                if (g == 0) pack(1, 1);
            }
            STAMP(4);
        };
        i32x4 qaA[KS], qaB[KS];
        f32x4 lsA[4], lsB[4];
        fetch_s_ops(1, qaA, lsA);
        STAMP(0);
        piece(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, qaA, lsA, qaB, lsB);
        piece(std::integral_constant<int, 0>{}, std::integral_constant<int, -1>{}, qaB, lsB, qaA, lsA);

        if (has_next) store_rowc(buf ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the next slice is in LDS
        STAMP(5);
        __syncthreads();
        STAMP(6);
        buf ^= 1;
        cur_h = nxt_h;
        cur_m = nxt_m;
        advance(nxt_h, nxt_m);
    }

#ifdef VFA_STAMP
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long *dst = g_stamps_w64 + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 7; i++) dst[i] = acc_[i];
        dst[7] = (unsigned long long)total_steps;
    }
#endif
    // ---- write dK (scaled) and dV of this wave's 64 keys, through LDS as whole rows, 32 keys at a time ----
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMA results have landed before they are read
    __syncthreads();
    char *ob = smem + wave * 32 * Cfg::OUT_ROWB;
    unsigned short *dkg = (unsigned short *)p.dk + bi.k_off(p.dk_bs, p.dk_rs, bidb) + (int64_t)hk * p.dk_hs;
    unsigned short *dvg = (unsigned short *)p.dv + bi.k_off(p.dv_bs, p.dv_rs, bidb) + (int64_t)hk * p.dv_hs;
#pragma unroll
    for (int which = 0; which < 2; which++) {
#pragma unroll
        for (int kh = 0; kh < 2; kh++) {
#pragma unroll
            for (int db = 0; db < NDB; db++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    i32x2 w2;
                    if (which == 0) {
                        w2[0] = pack2<BF16>(dk[kh][db][4 * g4] * p.scale, dk[kh][db][4 * g4 + 1] * p.scale);
                        w2[1] = pack2<BF16>(dk[kh][db][4 * g4 + 2] * p.scale, dk[kh][db][4 * g4 + 3] * p.scale);
                    } else {
                        w2[0] = pack2<BF16>(dv[kh][db][4 * g4], dv[kh][db][4 * g4 + 1]);
                        w2[1] = pack2<BF16>(dv[kh][db][4 * g4 + 2], dv[kh][db][4 * g4 + 3]);
                    }
                    *(i32x2 *)(ob + r * Cfg::OUT_ROWB + (32 * db + 8 * g4 + 4 * hh) * 2) = w2;
                }
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's own LDS writes
            __builtin_amdgcn_wave_barrier();
            unsigned short *dst = which == 0 ? dkg : dvg;
            const int64_t rs = which == 0 ? p.dk_rs : p.dv_rs;
            const int kr0 = kw0 + 32 * kh;
#pragma unroll
            for (int i = 0; i < (32 * CPR) / 64; i++) {
                const int c = lane + 64 * i;
                const int row = c / CPR, ch = c % CPR;
                const i32x4 w = *(const i32x4 *)(ob + row * Cfg::OUT_ROWB + ch * 16);
                if (kr0 + row < sk && ch * 8 < p.d) *(i32x4 *)(dst + (int64_t)(kr0 + row) * rs + ch * 8) = w;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);  // the reads above are done before the staging rows are rewritten
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int DP, bool BF16>
static hipError_t launch_w64(const KParams &p, hipStream_t stream) {
    using Cfg = Bwd64Cfg<DP>;
    const int n_kblocks = (p.seqlen_k + Cfg::KB - 1) / Cfg::KB;
    const long long blocks = (long long)n_kblocks * p.b * p.h_k;
    if (blocks <= 0) return hipSuccess;
    auto kern = bwd_dkdv_w64_kernel<DP, BF16>;
    static std::atomic<uint64_t> lds_set{0};
    if (hipError_t e = set_dyn_lds_once((const void *)kern, Cfg::LDS_BYTES, lds_set); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p, n_kblocks);
    return hipGetLastError();
}

// The 64-keys-per-wave dK/dV kernel: d <= 128 (DP 64 / 128), no dropout, no head split; the caller
// has checked that the Q / dO extents fit a 32-bit buffer offset.
hipError_t run_bwd_dkdv_w64(const KParams &p, hipStream_t stream) {
    const bool bf = p.is_bf16;
    if (p.d <= 64) return bf ? launch_w64<64, true>(p, stream) : launch_w64<64, false>(p, stream);
    return bf ? launch_w64<128, true>(p, stream) : launch_w64<128, false>(p, stream);
}

}  // namespace vfa

#ifdef VFA_STAMP
extern "C" __attribute__((visibility("default"))) int vfa_debug_read_stamps_w64(unsigned long long *host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vfa::g_stamps_w64), n * sizeof(unsigned long long));
}
#endif
