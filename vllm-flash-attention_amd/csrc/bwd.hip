// FlashAttention-2 backward for gfx950 / MI355X.
//
// The reference fork ships no backward kernel (setup.py:154-169, flash_api.cpp:943-944); the
// boundary is its Python call sites (vllm_flash_attn/flash_attn_interface.py:135-154, :183-207)
// and the parameter block csrc/flash_attn/src/flash.h:145-183 (dq_accum, dsoftmax_sum, ...).
// Math (standard FA-2): with S = Q K^T (+ALiBi, masks) and the forward's LSE,
//   P = exp2(S*scale_log2 - LSE*log2e),  dV = P16^T dO,  dP = dO V^T,  delta = rowsum(dO * O),
//   dS = P * (dP - delta),  dK = scale * dS16^T Q,  dQ = scale * dS16 K.
//
// Three kernels:
//   bwd_preprocess : delta (the softmax_d output), one wave per query row group
//   bwd_main       : one workgroup (4 waves) per 128-key block of one (batch, KV head); each wave
//                    owns 32 keys and keeps their dK^T / dV^T in accumulator registers while the
//                    workgroup sweeps the query heads of the GQA group x 32-row query slices, so
//                    dK / dV need no reduction across workgroups.  S and dP are computed with the
//                    KEY on the MFMA lane, so P and dS are already B operands of the dV^T / dK^T
//                    products; only dS crosses LDS (transposed) for dQ, which each wave computes
//                    for one 32-wide d slice over all 128 keys and adds to the fp32 dq_accum with
//                    float atomics (128 contiguous bytes per row segment).
//   bwd_convert_dq : dq = round16(scale * dq_accum)
#include "vfa_kernel_params.h"

namespace vfa {

constexpr int BW_NW = 4;
constexpr int BW_NT = BW_NW * 64;
constexpr int BW_KB = 32 * BW_NW;  // keys per workgroup
constexpr int BW_QM = 32;          // query rows per step

template <bool BF16>
__global__ void __launch_bounds__(256)
bwd_preprocess_kernel(const KParams p) {
    // one 16-lane group per query row: delta = sum_d dO * O (fp32); also clears the row's fp32 dQ
    // accumulator so that callers need not pre-zero the scratch
    const int lane16 = threadIdx.x & 15;
    const int64_t row_id = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t total = (int64_t)p.b * p.h * p.seqlen_q;
    if (row_id >= total) return;
    const int sqi = (int)(row_id % p.seqlen_q);
    const int bh = (int)(row_id / p.seqlen_q);
    const int bidb = bh / p.h, bidh = bh % p.h;
    const BlockInfo bi(p, bidb);
    float acc = 0.f;
    if (sqi < bi.sq) {
        const unsigned short *o = (const unsigned short *)p.o + bi.q_off(p.o_bs, p.o_rs, bidb) +
                                  (int64_t)sqi * p.o_rs + (int64_t)bidh * p.o_hs;
        const unsigned short *g = (const unsigned short *)p.dout + bi.q_off(p.do_bs, p.do_rs, bidb) +
                                  (int64_t)sqi * p.do_rs + (int64_t)bidh * p.do_hs;
        const int64_t row_base = bi.sum_s_q < 0 ? (int64_t)bidb * p.seqlen_q : (int64_t)bi.sum_s_q;
        float *dqa = p.dq_accum + ((row_base + sqi) * p.h + bidh) * (int64_t)p.d;
        for (int c = lane16 * 8; c < p.d; c += 128) {
            *(f32x4 *)(dqa + c) = f32x4{0, 0, 0, 0};
            *(f32x4 *)(dqa + c + 4) = f32x4{0, 0, 0, 0};
            const i32x4 a = *(const i32x4 *)(o + c), b = *(const i32x4 *)(g + c);
#pragma unroll
            for (int w = 0; w < 4; w++) {
                acc += to_f32<BF16>((unsigned short)(a[w] & 0xffff)) * to_f32<BF16>((unsigned short)(b[w] & 0xffff));
                acc += to_f32<BF16>((unsigned short)((unsigned)a[w] >> 16)) * to_f32<BF16>((unsigned short)((unsigned)b[w] >> 16));
            }
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
    if (lane16 == 0) p.dsoftmax_sum[(int64_t)bh * p.seqlen_q_rounded + sqi] = acc;
}

template <int DP>
struct BwdCfg {
    static constexpr int CPR = DP / 8;
    static constexpr int KV_BYTES = BW_KB * DP * 2;          // one of K / V
    static constexpr int QT_BYTES = BW_QM * DP * 2;          // one of Q / dO slice
    static constexpr int QBUF_BYTES = 2 * QT_BYTES + 256;    // Q, dO, lse[32], delta[32]
    static constexpr int DS_BYTES = BW_KB * 64;              // dS^T image [key][32 q] 16-bit
    static constexpr int OFF_K = 0;
    static constexpr int OFF_V = KV_BYTES;
    static constexpr int OFF_Q = KV_BYTES;                   // two Q buffers; they reuse the V tile's
                                                             // space once V is in registers
    static constexpr int OFF_DS = OFF_Q + (2 * QBUF_BYTES > KV_BYTES ? 2 * QBUF_BYTES : KV_BYTES);
    static constexpr int MAIN_LDS = OFF_DS + DS_BYTES;
    static constexpr int OUT_ROWB = DP * 2 + 16;
    static constexpr int OUT_LDS = BW_NW * 32 * OUT_ROWB;
    static constexpr int LDS_BYTES = MAIN_LDS > OUT_LDS ? MAIN_LDS : OUT_LDS;
    static constexpr int KV_CH = BW_KB * CPR / BW_NT;        // chunks per thread per K (or V) tile
    static constexpr int Q_CH = BW_QM * CPR / BW_NT;         // chunks per thread per Q (or dO) slice
    static_assert(BW_QM * CPR % BW_NT == 0 || BW_QM * CPR < BW_NT, "slice must divide over threads");
};

template <int DP, bool BF16>
__global__ void __launch_bounds__(BW_NT, 1)
bwd_main_kernel(const KParams p, const int n_kblocks) {
    using Cfg = BwdCfg<DP>;
    constexpr int CPR = Cfg::CPR;
    constexpr int KS = DP / 16, NDB = DP / 32;
    constexpr int QCH = (BW_QM * CPR + BW_NT - 1) / BW_NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;

    // block -> (batch, kv head, key block); low key blocks (most query rows under a causal mask)
    // are dispatched first
    int L = blockIdx.x;
    const int n_block = L % n_kblocks; L /= n_kblocks;
    const int hk = L % p.h_k;
    const int bidb = L / p.h_k;

    const BlockInfo bi(p, bidb);
    const int sq = bi.sq, sk = bi.sk;
    const int k0 = n_block * BW_KB;
    if (k0 >= sk) return;
    const int kw0 = k0 + 32 * wave;  // first key of this wave

    // query-slice range that can see this key block
    int m_min = 0, m_max = (sq + BW_QM - 1) / BW_QM;
    if (p.is_causal || p.is_local) m_min = max(0, (k0 + sq - sk - p.wr) / BW_QM);
    if (p.is_local) m_max = min(m_max, (k0 + BW_KB + sq - sk + p.wl + BW_QM - 1) / BW_QM);

    const unsigned short *kg = (const unsigned short *)p.k + bi.k_off(p.k_bs, p.k_rs, bidb) + (int64_t)hk * p.k_hs;
    const unsigned short *vg = (const unsigned short *)p.v + bi.k_off(p.v_bs, p.v_rs, bidb) + (int64_t)hk * p.v_hs;

    // ---- stage K and V (128 keys) into LDS; zero rows past sk ----
    {
#pragma unroll
        for (int i = 0; i < Cfg::KV_CH; i++) {
            const int c = tid + i * BW_NT;
            const int row = c / CPR, ch = c % CPR;
            const bool ok = (k0 + row) < sk && ch * 8 < p.d;
            const i32x4 kk = ok ? *(const i32x4 *)(kg + (int64_t)(k0 + row) * p.k_rs + ch * 8) : i32x4{0, 0, 0, 0};
            const i32x4 vv = ok ? *(const i32x4 *)(vg + (int64_t)(k0 + row) * p.v_rs + ch * 8) : i32x4{0, 0, 0, 0};
            const int off = lds_off<DP>(row, ch);
            *(i32x4 *)(smem + Cfg::OFF_K + off) = kk;
            *(i32x4 *)(smem + Cfg::OFF_V + off) = vv;
        }
    }
    __syncthreads();
    // B-operand fragments of this wave's 32 keys: lane (r, hh) holds K[kw0 + r][16 s + 8 hh ..]
    i32x4 kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) {
        kf[s] = *(const i32x4 *)(smem + Cfg::OFF_K + lds_off<DP>(32 * wave + r, 2 * s + hh));
        vf[s] = *(const i32x4 *)(smem + Cfg::OFF_V + lds_off<DP>(32 * wave + r, 2 * s + hh));
    }

    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int i = 0; i < NDB; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) { dk[i][j] = 0.f; dv[i][j] = 0.f; }

    const float c2 = p.scale_log2;
    const bool has_alibi = p.alibi_slopes != nullptr;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
    const int n_steps_per_head = max(m_max - m_min, 0);
    const int total_steps = n_steps_per_head * p.h_ratio;

    // ---- Q / dO slice staging (register prefetch, two LDS buffers) ----
    i32x4 qreg[QCH], doreg[QCH];
    float rowc = 0.f;  // lse (tid < 32) or delta (32 <= tid < 64) of the slice
    auto head_of = [&](int step) { return hk * p.h_ratio + step / n_steps_per_head; };
    auto m_of = [&](int step) { return m_min + step % n_steps_per_head; };
    auto load_slice = [&](int step) {
        const int bidh = head_of(step), q0 = m_of(step) * BW_QM;
        const unsigned short *qg = (const unsigned short *)p.q + bi.q_off(p.q_bs, p.q_rs, bidb) + (int64_t)bidh * p.q_hs;
        const unsigned short *dog = (const unsigned short *)p.dout + bi.q_off(p.do_bs, p.do_rs, bidb) + (int64_t)bidh * p.do_hs;
#pragma unroll
        for (int i = 0; i < QCH; i++) {
            const int c = tid + i * BW_NT;
            const int row = c / CPR, ch = c % CPR;
            const bool ok = c < BW_QM * CPR && (q0 + row) < sq && ch * 8 < p.d;
            qreg[i] = ok ? *(const i32x4 *)(qg + (int64_t)(q0 + row) * p.q_rs + ch * 8) : i32x4{0, 0, 0, 0};
            doreg[i] = ok ? *(const i32x4 *)(dog + (int64_t)(q0 + row) * p.do_rs + ch * 8) : i32x4{0, 0, 0, 0};
        }
        if (tid < 64) {
            const int row = q0 + (tid & 31);
            const bool ok = row < sq;
            if (tid < 32) rowc = ok ? p.lse[((int64_t)bidb * p.h + bidh) * p.seqlen_q + row] : INFINITY;
            else rowc = ok ? p.dsoftmax_sum[((int64_t)bidb * p.h + bidh) * p.seqlen_q_rounded + row] : 0.f;
        }
    };
    auto store_slice = [&](int buf) {
        char *qb = smem + Cfg::OFF_Q + buf * Cfg::QBUF_BYTES;
#pragma unroll
        for (int i = 0; i < QCH; i++) {
            const int c = tid + i * BW_NT;
            if (c < BW_QM * CPR) {
                const int off = lds_off<DP>(c / CPR, c % CPR);
                *(i32x4 *)(qb + off) = qreg[i];
                *(i32x4 *)(qb + Cfg::QT_BYTES + off) = doreg[i];
            }
        }
        if (tid < 64) ((float *)(qb + 2 * Cfg::QT_BYTES))[tid] = rowc;
    };

    if (total_steps > 0) load_slice(0);
    __syncthreads();  // every wave has its V fragments: the V tile's space becomes the Q buffers
    if (total_steps > 0) store_slice(0);
    __syncthreads();

    float *dqa_base = p.dq_accum;
    int buf = 0;
    for (int step = 0; step < total_steps; step++) {
        const int bidh = head_of(step), q0 = m_of(step) * BW_QM;
        const bool has_next = step + 1 < total_steps;
        if (has_next) load_slice(step + 1);

        const char *qb = smem + Cfg::OFF_Q + buf * Cfg::QBUF_BYTES;
        const char *dob = qb + Cfg::QT_BYTES;
        const float *lse_s = (const float *)(qb + 2 * Cfg::QT_BYTES);
        const float *dl_s = lse_s + 32;

        // ---- S[q][key] = Q K^T, dP'[q][key] = dO V^T - delta[q]; key on the lane ----
        f32x16 s, dp;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const f32x4 dl = *(const f32x4 *)(dl_s + 8 * g4 + 4 * hh);
#pragma unroll
            for (int t = 0; t < 4; t++) { s[4 * g4 + t] = 0.f; dp[4 * g4 + t] = -dl[t]; }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const i32x4 qa = *(const i32x4 *)(qb + lds_off<DP>(r, 2 * ks + hh));
            s = mma32<BF16>(qa, kf[ks], s);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const i32x4 da = *(const i32x4 *)(dob + lds_off<DP>(r, 2 * ks + hh));
            dp = mma32<BF16>(da, vf[ks], dp);
        }

        // ---- P = exp2(S*c - LSE*log2e) under the forward's mask and bias; dS = P * dP' ----
        {
            const int col = kw0 + r;
            float alibi = 0.f;
            if (has_alibi) alibi = p.alibi_slopes[(int64_t)bidb * p.alibi_bs + bidh] / p.scale;
            const bool masked_tile = has_alibi || (kw0 + 32 > sk) || (q0 + 32 > sq) ||
                                     ((p.is_causal || p.is_local) && (kw0 + 32 > q0 + 1 + sk - sq + p.wr)) ||
                                     (p.is_local && (kw0 < q0 + 31 + sk - sq - p.wl));
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const f32x4 ls = *(const f32x4 *)(lse_s + 8 * g4 + 4 * hh);
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int j = 4 * g4 + t;
                    float x = s[j];
                    bool vis = true;
                    if (masked_tile) {
                        const int row = q0 + 8 * g4 + 4 * hh + t;
                        if (has_alibi)
                            x = p.is_causal ? x + alibi * (float)col
                                            : x - alibi * fabsf((float)(row + sk - sq - col));
                        const int lo = p.is_local ? max(0, row + sk - sq - p.wl) : 0;
                        const int hi = (p.is_causal || p.is_local) ? min(sk, row + 1 + sk - sq + p.wr) : sk;
                        vis = col >= lo && col < hi && row < sq;
                    }
                    const float pe = vis ? __builtin_amdgcn_exp2f(__builtin_fmaf(x, c2, -ls[t] * (float)M_LOG2E)) : 0.f;
                    s[j] = pe;
                    dp[j] = pe * dp[j];
                }
            }
        }
        // 16-bit B operands: k-step ks2 (16 query rows) = regs 8 ks2 .. 8 ks2 + 7
        i32x4 pf[2], dsf[2];
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
            for (int w = 0; w < 4; w++) {
                pf[ks2][w] = pack2<BF16>(s[8 * ks2 + 2 * w], s[8 * ks2 + 2 * w + 1]);
                dsf[ks2][w] = pack2<BF16>(dp[8 * ks2 + 2 * w], dp[8 * ks2 + 2 * w + 1]);
            }
        // dS^T image for dQ: lane (key r) stores 4 consecutive query rows (8 bytes) at [key][8 g4 + 4 hh]
        {
            char *dsb = smem + Cfg::OFF_DS + (32 * wave + r) * 64;
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                i32x2 w2;
                w2[0] = dsf[g4 >> 1][2 * (g4 & 1)];
                w2[1] = dsf[g4 >> 1][2 * (g4 & 1) + 1];
                *(i32x2 *)(dsb + (8 * g4 + 4 * hh) * 2) = w2;
            }
        }

        // ---- dV^T += dO^T P,  dK^T += Q^T dS  (A operands by transposed reads of the slice) ----
#pragma unroll
        for (int db = 0; db < NDB; db++) {
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ks2++) {
                const int row0 = 16 * ks2 + 4 * hh + tr_q;
                const int ch = 4 * db + 2 * tr_g + (tr_p >> 1);
                const int o0 = lds_off<DP>(row0, ch) + 8 * (tr_p & 1);
                const int o1 = lds_off<DP>(row0 + 8, ch) + 8 * (tr_p & 1);
                const i32x2 a0 = __builtin_bit_cast(i32x2, lds_read_tr16(dob + o0));
                const i32x2 a1 = __builtin_bit_cast(i32x2, lds_read_tr16(dob + o1));
                const i32x2 b0 = __builtin_bit_cast(i32x2, lds_read_tr16(qb + o0));
                const i32x2 b1 = __builtin_bit_cast(i32x2, lds_read_tr16(qb + o1));
                const i32x4 dot = {a0[0], a0[1], a1[0], a1[1]};
                const i32x4 qt = {b0[0], b0[1], b1[0], b1[1]};
                dv[db] = mma32<BF16>(dot, pf[ks2], dv[db]);
                dk[db] = mma32<BF16>(qt, dsf[ks2], dk[db]);
            }
        }
        __syncthreads();  // dS^T of all four waves is in LDS

        // ---- dQ[32 q][32 d slice] = dS[32 q][128 keys] K[128 keys][d slice]; d slices round-robin
        // over the waves ----
        for (int db = wave; db < NDB; db += BW_NW) {
            f32x16 dq;
#pragma unroll
            for (int j = 0; j < 16; j++) dq[j] = 0.f;
#pragma unroll
            for (int ks = 0; ks < BW_KB / 16; ks++) {
                const int key0 = 16 * ks + 4 * hh + tr_q;
                // A = dS (32 q x 16 keys): transposed read of the [key][q] image, 16 q per lane group
                const char *dsb = smem + Cfg::OFF_DS;
                const int qo = (16 * tr_g + 4 * tr_p) * 2;
                const i32x2 a0 = __builtin_bit_cast(i32x2, lds_read_tr16(dsb + key0 * 64 + qo));
                const i32x2 a1 = __builtin_bit_cast(i32x2, lds_read_tr16(dsb + (key0 + 8) * 64 + qo));
                // B = K (16 keys x 32 d): transposed read of the K tile
                const int ch = 4 * db + 2 * tr_g + (tr_p >> 1);
                const i32x2 b0 = __builtin_bit_cast(i32x2, lds_read_tr16(smem + Cfg::OFF_K + lds_off<DP>(key0, ch) + 8 * (tr_p & 1)));
                const i32x2 b1 = __builtin_bit_cast(i32x2, lds_read_tr16(smem + Cfg::OFF_K + lds_off<DP>(key0 + 8, ch) + 8 * (tr_p & 1)));
                const i32x4 av = {a0[0], a0[1], a1[0], a1[1]};
                const i32x4 bv = {b0[0], b0[1], b1[0], b1[1]};
                dq = mma32<BF16>(av, bv, dq);
            }
            // fp32 atomics: one register = two 128-byte row segments (rows differ by 4 between halves)
            const int64_t row_base = (bi.sum_s_q < 0 ? (int64_t)bidb * p.seqlen_q : (int64_t)bi.sum_s_q);
            float *dst = dqa_base + ((row_base + q0) * p.h + bidh) * (int64_t)p.d + 32 * db + r;
            if (32 * db + r < p.d) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int row = (j & 3) + 8 * (j >> 2) + 4 * hh;
                    if (q0 + row < sq) unsafeAtomicAdd(dst + (int64_t)row * p.h * p.d, dq[j]);
                }
            }
        }

        if (has_next) store_slice(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- write dK (scaled) and dV for this wave's 32 keys, through LDS as whole rows ----
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
        __syncthreads();
        unsigned short *dst = which == 0 ? dkg : dvg;
        const int64_t rs = which == 0 ? p.dk_rs : p.dv_rs;
#pragma unroll
        for (int i = 0; i < (32 * CPR) / 64; i++) {
            const int c = lane + 64 * i;
            const int row = c / CPR, ch = c % CPR;
            const i32x4 w = *(const i32x4 *)(ob + row * Cfg::OUT_ROWB + ch * 16);
            if (kw0 + row < sk && ch * 8 < p.d) *(i32x4 *)(dst + (int64_t)(kw0 + row) * rs + ch * 8) = w;
        }
        __syncthreads();
    }
}

// dq = round16(scale * dq_accum); one thread per 8-wide chunk
template <bool BF16>
__global__ void __launch_bounds__(256)
bwd_convert_dq_kernel(const KParams p) {
    const int cpr = p.d / 8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)p.b * p.seqlen_q * p.h * cpr;
    if (idx >= total) return;
    const int ch = (int)(idx % cpr);
    int64_t t = idx / cpr;
    const int bidh = (int)(t % p.h); t /= p.h;
    const int sqi = (int)(t % p.seqlen_q);
    const int bidb = (int)(t / p.seqlen_q);
    const BlockInfo bi(p, bidb);
    if (sqi >= bi.sq) return;
    const int64_t row_base = bi.sum_s_q < 0 ? (int64_t)bidb * p.seqlen_q : (int64_t)bi.sum_s_q;
    const float *src = p.dq_accum + ((row_base + sqi) * p.h + bidh) * (int64_t)p.d + ch * 8;
    const f32x4 a0 = *(const f32x4 *)src * p.scale, a1 = *(const f32x4 *)(src + 4) * p.scale;
    unsigned short *dst = (unsigned short *)p.dq + bi.q_off(p.dq_bs, p.dq_rs, bidb) + (int64_t)sqi * p.dq_rs +
                          (int64_t)bidh * p.dq_hs + ch * 8;
    i32x4 w;
    w[0] = pack2<BF16>(a0[0], a0[1]); w[1] = pack2<BF16>(a0[2], a0[3]);
    w[2] = pack2<BF16>(a1[0], a1[1]); w[3] = pack2<BF16>(a1[2], a1[3]);
    *(i32x4 *)dst = w;
}

// dk / dv rows of key blocks that no workgroup visits cannot exist (every key block < sk has a
// workgroup); rows of keys no query can see get zeros from the zero-initialised accumulators.

template <int DP, bool BF16>
static hipError_t launch_bwd(const KParams &p, hipStream_t stream, hipEvent_t *ev) {
    using Cfg = BwdCfg<DP>;
    hipError_t e;
    if (ev) (void)hipEventRecord(ev[0], stream);
    {
        const int64_t rows = (int64_t)p.b * p.h * p.seqlen_q;
        hipLaunchKernelGGL(bwd_preprocess_kernel<BF16>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, p);
    }
    const int n_kblocks = (p.seqlen_k + BW_KB - 1) / BW_KB;
    const long long blocks = (long long)n_kblocks * p.b * p.h_k;
    if (ev) (void)hipEventRecord(ev[1], stream);
    if (blocks > 0) {
        auto kern = bwd_main_kernel<DP, BF16>;
        if (Cfg::LDS_BYTES > 64 * 1024) {
            e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BW_NT), Cfg::LDS_BYTES, stream, p, n_kblocks);
    }
    if (ev) (void)hipEventRecord(ev[2], stream);
    {
        const int64_t items = (int64_t)p.b * p.seqlen_q * p.h * (p.d / 8);
        hipLaunchKernelGGL(bwd_convert_dq_kernel<BF16>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, p);
    }
    if (ev) (void)hipEventRecord(ev[3], stream);
    return hipGetLastError();
}

hipError_t run_bwd(const KParams &p, hipStream_t stream, hipEvent_t *ev) {
    const bool bf = p.is_bf16;
    if (p.d <= 64) return bf ? launch_bwd<64, true>(p, stream, ev) : launch_bwd<64, false>(p, stream, ev);
    if (p.d <= 128) return bf ? launch_bwd<128, true>(p, stream, ev) : launch_bwd<128, false>(p, stream, ev);
    return bf ? launch_bwd<256, true>(p, stream, ev) : launch_bwd<256, false>(p, stream, ev);
}

}  // namespace vfa
