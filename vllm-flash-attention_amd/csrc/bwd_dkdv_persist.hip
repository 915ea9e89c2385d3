// FlashAttention-2 backward, dK/dV kernel with PERSISTENT workgroups (gfx950 / MI355X).
//
// The key-stationary kernel of bwd.hip (bwd_dkdv_kernel, 8 waves x 32 keys: K fragments and the
// dK^T / dV^T accumulators in registers, the V tile resident in LDS, 64-row Q / dO slices of the
// group's query heads streaming through two LDS buffers; S and dP with the key on the lane, P and
// dS directly the B operands of dV^T += dO^T P and dK^T += Q^T dS) with one workgroup per CU that
// walks a list of (batch, kv head, 256-key block) items.  The reference ships no backward kernel
// (setup.py:154-169); the boundary is vllm_flash_attn/flash_attn_interface.py:135-154.
//
// One block per workgroup spends 6.2 us before its first step (kernel arguments, block mapping, one
// memory round trip for K, V and the first slice) and 2.3 us after its last (dK / dV through LDS to
// memory) with nothing to overlap them: 8.5 us against a 144 us step loop at seqlen 4096, 10 us
// against 23-34 us at 512.  Here the slices of consecutive blocks form one stream: the first slice
// of the next block is requested under the last step of this one (into the idle slice buffer), and
// at the seam the next block's K fragments (registers) and V tile (straight into LDS) are requested
// BEFORE this block's dK / dV are staged (through the slice buffer that has just been freed) and
// stored, so their round trip runs under the epilogue.
//
// Work distribution: static, as in fwd_prefill_persist.hip; the launch order is that of
// bwd_dkdv_kernel (whole (batch, kv head) pairs per XCD, key blocks heaviest first across groups
// of pairs).
//
// Scope: dense batches, 64 < d <= 128, causal or full attention, no ALiBi / window / dropout, no
// head split.
#include "vfa_kernel_params.h"
#include <type_traits>

namespace vfa {

struct DkdvPersistCfg {
    static constexpr int DP = 128, NW = 8, NT = NW * 64, KB = NW * 32, CPR = DP / 8;
    static constexpr int QM = 64;                                // query rows per step: two 32-row sub-blocks
    static constexpr int QT_BYTES = QM * DP * 2;                 // one of Q / dO slice
    static constexpr int QBUF_BYTES = 2 * QT_BYTES + 8 * QM;     // Q, dO, -lse/scale[QM], -delta[QM]
    static constexpr int OFF_VT = 2 * QBUF_BYTES;                // V tile behind the two slice buffers
    static constexpr int KV_BYTES = KB * DP * 2;
    static constexpr int LDS_BYTES = OFF_VT + KV_BYTES;
    static constexpr int OST_BYTES = 32 * 128;                   // a wave's dK / dV rows, one 64-column half at a time
    static_assert(NW * OST_BYTES <= QBUF_BYTES, "dK / dV staging fits a slice buffer");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per workgroup");
};

template <bool BF16>
__global__ void __launch_bounds__(DkdvPersistCfg::NT, 1)
bwd_dkdv_persist_kernel(const KParams p, const int n_kblocks, const int n_items) {
    using Cfg = DkdvPersistCfg;
    constexpr int DP = Cfg::DP, NW = Cfg::NW, NT = Cfg::NT, KB = Cfg::KB, CPR = Cfg::CPR, QM = Cfg::QM;
    constexpr int KS = DP / 16, NDB = DP / 32;
    constexpr int RPI = 64 / CPR;                       // rows per wave instruction of a direct-to-LDS load
    constexpr int NIQ = QM / (RPI * NW);                // instructions per wave, slice and tensor
    constexpr int NIV = KB / (RPI * NW);                // instructions per wave and V tile
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int r_ = r, hh_ = hh, lane_ = lane, tid_ = tid;   // (shadowed by opaque copies in the seam code)
    const int sq = p.seqlen_q, sk = p.seqlen_k;         // dense batch
    const float c2 = p.scale_log2;
    const int nh = p.h_ratio;                           // query heads per kv head: all swept by the block

    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwx = gridDim.x >> 3;
    const int n_x = (n_items - xcd + 7) >> 3;
    const int rounds = (n_x + nwx - 1) / nwx;

    struct Blk { int bidb, hk, k0, m_min, m_max; };     // 64-row slices [m_min, m_max) see the block's keys
    // This workgroup's next block with rows that see it; a block no row sees gets dK = dV = 0 here.
    auto next_item = [&](int &t, Blk &b) __attribute__((always_inline)) -> bool {
        for (;;) {
            if (++t >= rounds) return false;
            const int s = t * nwx + ((t & 1) ? nwx - 1 - wg : wg);
            if (s >= n_x) continue;
            // launch order -> (pair, key block): see bwd_dkdv_kernel
            const int P = p.b * p.h_k, L = 8 * s + xcd;
            const int full = (P / 8) * 8 * n_kblocks;
            int pair, n_block;
            if (L < full) {
                const int ppx = P / 8;
                const int G = max(1, min(ppx, (64 + n_kblocks - 1) / n_kblocks));
                const int g = s / (G * n_kblocks);
                const int within = s - g * G * n_kblocks;
                const int Gl = min(G, ppx - g * G);
                n_block = within / Gl;
                pair = (g * G + within % Gl) * 8 + xcd;
            } else {
                const int Lr = L - full, R = P - (P / 8) * 8;
                pair = (P / 8) * 8 + Lr % R;
                n_block = Lr / R;
            }
            b.hk = pair % p.h_k;
            b.bidb = pair / p.h_k;
            b.k0 = n_block * KB;
            b.m_min = p.is_causal ? max(0, (b.k0 + sq - sk) / QM) : 0;
            b.m_max = (sq + QM - 1) / QM;
            if (b.m_max > b.m_min) return true;
            unsigned short *dkg = (unsigned short *)p.dk + (int64_t)b.bidb * p.dk_bs + (int64_t)b.hk * p.dk_hs;
            unsigned short *dvg = (unsigned short *)p.dv + (int64_t)b.bidb * p.dv_bs + (int64_t)b.hk * p.dv_hs;
            int tid = tid_;   // (opaque: see the epilogue)
            asm volatile("" : "+v"(tid));
            for (int c = tid; c < KB * CPR; c += NT) {
                const int row = b.k0 + c / CPR, ch = c % CPR;
                if (row < sk && ch * 8 < p.d) {
                    *(i32x4 *)(dkg + (int64_t)row * p.dk_rs + ch * 8) = i32x4{0, 0, 0, 0};
                    *(i32x4 *)(dvg + (int64_t)row * p.dv_rs + ch * 8) = i32x4{0, 0, 0, 0};
                }
            }
        }
    };

    // ---- per-lane offsets of the direct-to-LDS loads: lane -> (row, chunk) of its slot in a wave
    // instruction's 1 KiB; the lane fetches the chunk the XOR swizzle places there (the rows of a
    // wave's instructions are 32 apart: same swizzle) ----
    const unsigned q_rs2 = (unsigned)p.q_rs * 2, do_rs2 = (unsigned)p.do_rs * 2, v_rs2 = (unsigned)p.v_rs * 2;
    unsigned dq_voff, ddo_voff;
    const int ld_row0 = RPI * wave + lane / CPR;
    {
        const int ch = (lane % CPR) ^ swz<DP>(ld_row0);
        const bool ok = ch * 8 < p.d;
        dq_voff = ok ? (unsigned)(ld_row0 * (int)q_rs2 + ch * 16) : 0x80000000u;
        ddo_voff = ok ? (unsigned)(ld_row0 * (int)do_rs2 + ch * 16) : 0x80000000u;
    }
    const unsigned q_bytes = (unsigned)(((int64_t)(sq - 1) * p.q_rs + p.d) * 2);
    const unsigned do_bytes = (unsigned)(((int64_t)(sq - 1) * p.do_rs + p.d) * 2);
    const unsigned k_bytes = (unsigned)(((int64_t)(sk - 1) * p.k_rs + p.d) * 2);
    const unsigned v_bytes = (unsigned)(((int64_t)(sk - 1) * p.v_rs + p.d) * 2);

    // Q / dO slice m of query head gh of block b, straight into slice buffer `buf`
    auto load_slice = [&](const Blk &b, const int gh, const int m, const int buf) __attribute__((always_inline)) {
        const int bidh = b.hk * nh + gh;
        const unsigned short *qg = (const unsigned short *)p.q + (int64_t)b.bidb * p.q_bs + (int64_t)bidh * p.q_hs;
        const unsigned short *dog = (const unsigned short *)p.dout + (int64_t)b.bidb * p.do_bs + (int64_t)bidh * p.do_hs;
        const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)qg, 0, q_bytes, 0x00020000);
        const auto do_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dog, 0, do_bytes, 0x00020000);
        char *qb = smem + buf * Cfg::QBUF_BYTES;
#pragma unroll
        for (int i = 0; i < NIQ; i++) {
            const unsigned rows = (unsigned)(m * QM + RPI * NW * i);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rsrc, (lds_ptr)(qb + (wave + NW * i) * 1024), 16,
                                                     dq_voff + rows * q_rs2, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(do_rsrc, (lds_ptr)(qb + Cfg::QT_BYTES + (wave + NW * i) * 1024), 16,
                                                     ddo_voff + rows * do_rs2, 0, 0, 0);
        }
    };
    // LSE (wave 0) and delta (wave 1) of the slice's 64 rows, one per lane; +inf / 0 past the last
    // row.  (The choice of array is wave-uniform on purpose: as a per-lane select the compiler reads
    // the array's ADDRESS out of the kernel arguments with a vector load per step.)
    static_assert(QM == 64, "one wave per row-constant array");
    float rowc = 0.f;
    auto load_rowc = [&](const Blk &b, const int gh, const int m) __attribute__((always_inline)) {
        const int bidh = b.hk * nh + gh;
        const int row = m * QM + lane;
        if (wave == 0) {
            const float *src = p.lse + ((int64_t)b.bidb * p.h + bidh) * p.seqlen_q;
            rowc = row < sq ? src[row] : INFINITY;
        } else if (wave == 1) {
            const float *src = p.dsoftmax_sum + ((int64_t)b.bidb * p.h + bidh) * p.seqlen_q_rounded;
            rowc = row < sq ? src[row] : 0.f;
        }
    };
    // ... stored as what the S and dP accumulators start from: -LSE / scale and -delta
    auto store_rowc = [&](const int buf) __attribute__((always_inline)) {
        float *dst = (float *)(smem + buf * Cfg::QBUF_BYTES + 2 * Cfg::QT_BYTES);
        if (wave == 0) dst[lane] = -rowc / p.scale;
        else if (wave == 1) dst[QM + lane] = -rowc;
    };
    // K fragments of this wave's 32 keys (B operands of S = Q K^T; lane (r, hh) holds
    // K[kw0 + r][16 s + 8 hh ..]) straight from memory, and the block's V tile straight into LDS;
    // rows past sk and the head-dim padding read as zeros
    i32x4 kf[KS];
    auto load_kv = [&](const Blk &b) __attribute__((always_inline)) {
        const unsigned short *kg = (const unsigned short *)p.k + (int64_t)b.bidb * p.k_bs + (int64_t)b.hk * p.k_hs;
        const unsigned short *vg = (const unsigned short *)p.v + (int64_t)b.bidb * p.v_bs + (int64_t)b.hk * p.v_hs;
        const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)kg, 0, k_bytes, 0x00020000);
        const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)vg, 0, v_bytes, 0x00020000);
        int r = r_, hh = hh_, lane = lane_;   // (opaque: see the epilogue)
        asm volatile("" : "+v"(r), "+v"(hh), "+v"(lane));
        const unsigned krow = (unsigned)(b.k0 + 32 * wave + r) * (unsigned)p.k_rs * 2;
        const int vrow0 = RPI * wave + lane / CPR;
        const int vch = (lane % CPR) ^ swz<DP>(vrow0);
        const unsigned dv_voff = vch * 8 < p.d ? (unsigned)(vrow0 * (int)v_rs2 + vch * 16) : 0x80000000u;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int col = (2 * s + hh) * 8;
            kf[s] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                k_rsrc, col < p.d ? krow + (unsigned)col * 2 : 0x80000000u, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NIV; i++) {
            const unsigned rows = (unsigned)(b.k0 + RPI * NW * i);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_ptr)(smem + Cfg::OFF_VT + (wave + NW * i) * 1024), 16,
                                                     dv_voff + rows * v_rs2, 0, 0, 0);
        }
    };

    f32x16 dk[NDB], dv[NDB];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NDB; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) { dk[i][j] = 0.f; dv[i][j] = 0.f; }
    };
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
    const char *vt = smem + Cfg::OFF_VT;

    // dk = round16(scale * dK), dv = round16(dV) of block b, through `ost` (this wave's 4 KiB of the
    // slice buffer that has just been freed) as whole 128-byte half rows
    auto epilogue = [&](const Blk &b, char *ost) __attribute__((always_inline)) {
        unsigned short *dkg = (unsigned short *)p.dk + (int64_t)b.bidb * p.dk_bs + (int64_t)b.hk * p.dk_hs;
        unsigned short *dvg = (unsigned short *)p.dv + (int64_t)b.bidb * p.dv_bs + (int64_t)b.hk * p.dv_hs;
        const int kw0 = b.k0 + 32 * wave;
        // (opaque copies of the lane's coordinates: every per-lane address below is an invariant of
        //  the whole walk, and what the compiler hoists out of it stays in registers through the
        //  step loop, whose schedule degrades with every register it loses)
        int r = r_, hh = hh_, lane = lane_;
        asm volatile("" : "+v"(r), "+v"(hh), "+v"(lane));
        const int wsw = (r >> 1) & 7;   // chunk swizzle of the staging rows: see fwd_prefill_persist.hip
#pragma unroll
        for (int which = 0; which < 2; which++) {
            unsigned short *dst = which == 0 ? dkg : dvg;
            const int64_t rs = which == 0 ? p.dk_rs : p.dv_rs;
            const float sc = which == 0 ? p.scale_rp : 1.f;
#pragma unroll
            for (int half = 0; half < 2; half++) {
#pragma unroll
                for (int d2 = 0; d2 < 2; d2++)
#pragma unroll
                    for (int g4 = 0; g4 < 4; g4++) {
                        const int db = 2 * half + d2;
                        const f32x16 &a = which == 0 ? dk[db] : dv[db];
                        i32x2 w;
                        w[0] = pack2<BF16>(a[4 * g4 + 0] * sc, a[4 * g4 + 1] * sc);
                        w[1] = pack2<BF16>(a[4 * g4 + 2] * sc, a[4 * g4 + 3] * sc);
                        *(i32x2 *)(ost + r * 128 + (((4 * d2 + g4) ^ wsw) << 4) + 8 * hh) = w;
                    }
                // (the wave's own staging rows; a wave's LDS operations execute in order)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int row = (lane >> 3) + 8 * i, ch = lane & 7;
                    const i32x4 w = *(const i32x4 *)(ost + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
                    const int col = 64 * half + 8 * ch;
                    if (kw0 + row < sk && col < p.d) *(i32x4 *)(dst + (int64_t)(kw0 + row) * rs + col) = w;
                }
            }
        }
    };

    // ---- first block ----
    Blk cur, nxt;
    int t = -1;
    if (!next_item(t, cur)) return;
    load_kv(cur);
    int cur_h = 0, cur_m = cur.m_max - 1;
    load_slice(cur, cur_h, cur_m, 0);
    load_rowc(cur, cur_h, cur_m);
    bool have_next_block = next_item(t, nxt);
    zero_acc();
    store_rowc(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();

    // Two loops: the step loop of a block (the body of bwd_dkdv_kernel's, whose last step requests
    // the first slice of the next block instead of nothing) inside the walk over the blocks; what
    // the seam needs (K / V addressing, dK / dV stores, block mapping) stays outside the step loop
    // and out of its registers.
    auto advance = [&](int &hh_, int &m_) {
        if (++hh_ == nh) {
            hh_ = 0;
            --m_;
        }
    };
    int buf = 0;
    for (;;) {
    const int total_steps = (cur.m_max - cur.m_min) * nh;
    int nxt_h = cur_h, nxt_m = cur_m;
    advance(nxt_h, nxt_m);
    for (int step = 0; step < total_steps; step++) {
        const bool in_block = step + 1 < total_steps;
        const bool has_next = in_block || have_next_block;
        if (in_block) {
            load_slice(cur, nxt_h, nxt_m, buf ^ 1);
            load_rowc(cur, nxt_h, nxt_m);
        }
#ifndef EXP_NOSEAM
        else if (have_next_block) {
            // the seam's slice: first step of the next block, into the idle buffer
            load_slice(nxt, 0, nxt.m_max - 1, buf ^ 1);
            load_rowc(nxt, 0, nxt.m_max - 1);
        }
#endif

        const char *qb0 = smem + buf * Cfg::QBUF_BYTES;
        const float *lse_s0 = (const float *)(qb0 + 2 * Cfg::QT_BYTES);
        const int kw0 = cur.k0 + 32 * wave;   // first key of this wave
#pragma unroll
        for (int sub = 1; sub >= 0; sub--) {  // 32-row sub-blocks, last rows first
            const int q0 = cur_m * QM + 32 * sub;
            const char *qb = qb0 + 32 * sub * (DP * 2);
            const char *dob = qb + Cfg::QT_BYTES;
            const float *lse_s = lse_s0 + 32 * sub;
            const float *dl_s = lse_s0 + QM + 32 * sub;
            // a wave whose 32 keys are invisible to all 32 rows of this sub-block only helps with staging
            const bool skip = p.is_causal && kw0 >= q0 + 32 + sk - sq;
            if (!skip) {
                // ---- S[q][key] = Q K^T - LSE[q] / scale; key on the lane ----
                f32x16 s;
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const f32x4 ls = *(const f32x4 *)(lse_s + 8 * g4 + 4 * hh);
#pragma unroll
                    for (int e = 0; e < 4; e++) s[4 * g4 + e] = ls[e];
                }
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const i32x4 qa = *(const i32x4 *)(qb + lds_off<DP>(r, 2 * ks + hh));
                    s = mma32<BF16>(qa, kf[ks], s);
                }
#ifndef EXP_NOSCHED
                // order for the scheduler: the four row-constant reads and the fragments of two
                // k-steps first, then every MFMA with the read of the k-step two ahead behind it (left
                // alone it puts every read directly in front of its MFMA, which then waits for it)
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (ks + 2 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#endif
                // ---- P = exp2(S * c) under the forward's mask ----
                const int col = kw0 + r;
                const bool masked_tile = (kw0 + 32 > sk) || (q0 + 32 > sq) ||
                                         (p.is_causal && (kw0 + 32 > q0 + 1 + sk - sq));
                if (masked_tile) {
                    // register j of lane (r, hh) is local row roff(j) = (j & 3) + 8 (j >> 2) + 4 hh, the
                    // lane's key is `col`: visible iff roff >= col - shift (causal) and roff < sq - q0 - 4 hh
                    constexpr int BIG = 1 << 28;
                    const int shift = q0 + 4 * hh + sk - sq;
                    int lo_r = p.is_causal ? col - shift : -BIG;
                    int hi_r = min(BIG, sq - q0 - 4 * hh - 1);
                    lo_r = max(lo_r, -BIG);
                    if (col >= sk || hi_r < lo_r) { lo_r = BIG; hi_r = BIG; }   // nothing visible
                    const unsigned span = (unsigned)(hi_r - lo_r);
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const int roff = (j & 3) + 8 * (j >> 2);
                        s[j] = ((unsigned)(roff - lo_r) <= span) ? s[j] : -INFINITY;
                    }
                }
#pragma unroll
                for (int j = 0; j < 16; j++) s[j] = __builtin_amdgcn_exp2f(s[j] * c2);
                // ---- dP'[q][key] = dO V^T - delta[q];  dS = P * dP' ----
                f32x16 dp;
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const f32x4 dl = *(const f32x4 *)(dl_s + 8 * g4 + 4 * hh);
#pragma unroll
                    for (int e = 0; e < 4; e++) dp[4 * g4 + e] = dl[e];   // (-delta)
                }
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const i32x4 da = *(const i32x4 *)(dob + lds_off<DP>(r, 2 * ks + hh));
                    const i32x4 vb = *(const i32x4 *)(vt + lds_off<DP>(32 * wave + r, 2 * ks + hh));
                    dp = mma32<BF16>(da, vb, dp);
                }
#ifndef EXP_NOSCHED
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // row constants, fragments of two k-steps
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (ks + 2 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
#endif
                // 16-bit B operands: k-step ks2 (16 query rows) = regs 8 ks2 .. 8 ks2 + 7
                i32x4 pf[2], dsf[2];
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const int j = 8 * ks2 + 2 * w;
                        pf[ks2][w] = pack2<BF16>(s[j], s[j + 1]);
                        dsf[ks2][w] = pack2<BF16>(s[j] * dp[j], s[j + 1] * dp[j + 1]);
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
            }  // !skip
            __builtin_amdgcn_sched_barrier(0);  // keep the sub-blocks apart: interleaving them only adds live registers
        }  // sub

        if (has_next) store_rowc(buf ^ 1);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the next slice is in LDS
        __syncthreads();
        buf ^= 1;
        cur_h = nxt_h;
        cur_m = nxt_m;
        advance(nxt_h, nxt_m);
    }
    // ---- the seam: every wave is done with this block's K fragments, V tile and slices; the next
    // block's first slice sits in buffer `buf`, the other one is free for the staging ----
    if (have_next_block) load_kv(nxt);
    epilogue(cur, smem + (buf ^ 1) * Cfg::QBUF_BYTES + wave * Cfg::OST_BYTES);
    if (!have_next_block) break;
    zero_acc();
    cur = nxt;
    cur_h = 0;
    cur_m = cur.m_max - 1;
    have_next_block = next_item(t, nxt);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): K fragments and this wave's part of the V tile
    __syncthreads();
    }
}

bool bwd_dkdv_persist_ok(const KParams &p) {
    if (p.d <= 64 || p.d > 128) return false;
    if (p.cu_seqlens_q || p.cu_seqlens_k || p.seqused_k || p.block_table || p.cache_batch_idx) return false;
    if (p.alibi_slopes || p.is_local || p.hsplit > 1) return false;
    if (p.seqlen_q <= 0 || p.seqlen_k <= 0) return false;
    const int64_t lim = (int64_t)1 << 30;
    const int64_t rq = (int64_t)p.seqlen_q + 256, rk = (int64_t)p.seqlen_k + 256;
    return rq * p.q_rs * 2 < lim && rq * p.do_rs * 2 < lim && rk * p.k_rs * 2 < lim && rk * p.v_rs * 2 < lim;
}

long long bwd_dkdv_persist_items(const KParams &p) {
    return (long long)((p.seqlen_k + DkdvPersistCfg::KB - 1) / DkdvPersistCfg::KB) * p.b * p.h_k;
}

hipError_t run_bwd_dkdv_persist(const KParams &p, hipStream_t stream) {
    using Cfg = DkdvPersistCfg;
    const int n_kblocks = (p.seqlen_k + Cfg::KB - 1) / Cfg::KB;
    const long long items = bwd_dkdv_persist_items(p);
    if (items <= 0) return hipSuccess;
    if (items > 0x7fffffffLL) return hipErrorInvalidValue;
    const int grid = std::max(8, device_num_cus() / 8 * 8);
    static std::atomic<uint64_t> lds_set[2];
    hipError_t e;
    if (p.is_bf16) {
        auto kern = bwd_dkdv_persist_kernel<true>;
        if ((e = set_dyn_lds_once((const void *)kern, Cfg::LDS_BYTES, lds_set[0])) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p, n_kblocks, (int)items);
    } else {
        auto kern = bwd_dkdv_persist_kernel<false>;
        if ((e = set_dyn_lds_once((const void *)kern, Cfg::LDS_BYTES, lds_set[1])) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p, n_kblocks, (int)items);
    }
    return hipGetLastError();
}

}  // namespace vfa
