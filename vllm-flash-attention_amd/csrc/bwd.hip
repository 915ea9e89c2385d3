// FlashAttention-2 backward for gfx950 / MI355X.
//
// The reference fork ships no backward kernel (setup.py:154-169, flash_api.cpp:943-944); the
// boundary is its Python call sites (vllm_flash_attn/flash_attn_interface.py:135-154, :183-207)
// and the parameter block csrc/flash_attn/src/flash.h:145-183 (dq_accum, dsoftmax_sum, ...).
// Math (standard FA-2): with S = Q K^T (+ALiBi, masks) and the forward's LSE,
//   P = exp2(S*scale_log2 - LSE*log2e),  dV = P16^T dO,  dP = dO V^T,  delta = rowsum(dO * O),
//   dS = P * (dP - delta),  dK = scale * dS16^T Q,  dQ = scale * dS16 K.
// With dropout (template parameter DROP; keep mask M from the forward's Philox stream, keep
// probability pk; the upstream FlashAttention-2 formulation behind the rp_dropout /
// scale_softmax_rp_dropout fields, flash.h:109-117): dV = (P16 * M)^T dO / pk, delta is scaled by
// pk, dS = P * (M * dP - pk delta), and dQ, dK carry scale / pk.
//
// Three kernels, no atomics (dq, dk, dv are bitwise reproducible):
//   bwd_preprocess : delta (the softmax_d output), one 16-lane group per query row
//   bwd_dkdv       : KEY-stationary.  One workgroup (8 waves, or 4 for short key sequences) per
//                    256- (128-) key block of one (batch, KV head); each wave owns 32 keys, holds
//                    their K MFMA fragments and the dK^T / dV^T accumulators in registers, and the
//                    workgroup sweeps the query heads of the GQA group x 32- or 64-row query slices
//                    (staged through LDS), so dK / dV need no reduction across workgroups.  S and dP are computed with the key on the MFMA
//                    lane, so P and dS are directly the B operands of dV^T += dO^T P and
//                    dK^T += Q^T dS (A operands by transposed LDS reads of the slice).
//   bwd_dq         : QUERY-stationary, the forward kernel's structure: a wave owns 32 query rows
//                    (Q^T and dO^T fragments in registers, LSE and delta one scalar per lane), K/V
//                    tiles of 64 keys stream through LDS; S^T = K Q^T and dP^T = V dO^T put the
//                    query on the lane, dS^T is then the B operand of dQ^T += K^T dS^T (K^T by
//                    transposed reads).  Recomputing S and dP here costs 2 extra products (7 instead
//                    of 5) and removes the fp32 dQ atomics, which on this chip run at ~1.3 TB/s and
//                    were the bound of a one-kernel design.
#include "vfa_kernel_params.h"
#include "vfa_dispatch.h"
#include "vfa_one_wave_helpers.h"

namespace vfa {

#if defined(VFA_STAMP) && (defined(VFA_DROPOUT_TU) || defined(VFA_DQ_ILP_TU))
#undef VFA_STAMP
#endif
#ifdef VFA_STAMP
// Diagnostic build only: per-phase shader-clock sums of the dK/dV step loop.
__device__ unsigned long long g_stamps_dkdv[512 * 8 * 8];
#define STAMP(slot)                                                        \
    do {                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                 \
        unsigned long long t_;                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                 \
        acc_[slot] += t_ - last_;                                          \
        last_ = t_;                                                        \
    } while (0)
// ... and the wall-clock (100 MHz) times at which a workgroup enters the kernel, starts and ends its
// step loop and leaves (slots 0-3), and inside the prologue: block mapping done (4), first slice
// requested (5), own loads landed (6): the prologue / loop / epilogue split and the occupancy
// timeline of a launch.
__device__ unsigned long long g_blk_dkdv[4096 * 8];
__device__ unsigned long long g_blk_dq[4096 * 8];   // the same for the dQ kernel
#define BLK_STAMP_TO(arr, i)                                                                      \
    do {                                                                                          \
        if (wave == 0 && blockIdx.x < 4096 && blockIdx.y == 0 && blockIdx.z == 0) {               \
            unsigned long long t_;                                                                \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            if (lane == 0) arr[blockIdx.x * 8 + (i)] = t_;                                        \
        }                                                                                         \
    } while (0)
#define BLK_STAMP(i) BLK_STAMP_TO(g_blk_dkdv, i)
#define DQ_STAMP(i) BLK_STAMP_TO(g_blk_dq, i)
#else
#define STAMP(slot) do { } while (0)
#define BLK_STAMP(i) do { } while (0)
#define DQ_STAMP(i) do { } while (0)
#endif

#ifndef VFA_PP_PRIO
#define VFA_PP_PRIO 2
#endif
#ifndef VFA_DKDV_PP_DEFAULT
#define VFA_DKDV_PP_DEFAULT 0
#endif

// Scheduling-group directive with a computed (compile-time) size; a compile-time loop.
template <int MASK, int N>
__device__ __forceinline__ void sched_group() {
    if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Workgroup barrier that leaves this wave's global -> LDS loads in flight: the LDS reads and
// writes issued so far are complete (lgkmcnt), vmcnt is not touched.
__device__ __forceinline__ void phase_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Query rows per step of the dK/dV kernel: sub-blocks of 32 rows, QSUB of them per staged slice
// (one barrier and one load/store round per slice, so a taller slice amortises them).

template <bool BF16>
__global__ void __launch_bounds__(256)
bwd_preprocess_kernel(const KParams p) {
    // one 16-lane group per query row: delta = sum_d dO * O (fp32)
    const int lane16 = threadIdx.x & 15;
    // rows in (batch, query, head) order: the 16 rows of a workgroup are 16 heads of one query
    // position, contiguous in the usual (batch, seqlen, heads, d) layout
    const int64_t row_id = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t total = (int64_t)p.b * p.h * p.seqlen_q;
    if (row_id >= total) return;
    const int bidh = (int)(row_id % p.h);
    const int64_t bs = row_id / p.h;
    const int sqi = (int)(bs % p.seqlen_q);
    const int bidb = (int)(bs / p.seqlen_q);
    const int bh = bidb * p.h + bidh;
    const BlockInfo bi(p, bidb);
    float acc = 0.f;
    if (sqi < bi.sq) {
        const unsigned short *o = (const unsigned short *)p.o + bi.q_off(p.o_bs, p.o_rs, bidb) +
                                  (int64_t)sqi * p.o_rs + (int64_t)bidh * p.o_hs;
        const unsigned short *g = (const unsigned short *)p.dout + bi.q_off(p.do_bs, p.do_rs, bidb) +
                                  (int64_t)sqi * p.do_rs + (int64_t)bidh * p.do_hs;
        for (int c = lane16 * 8; c < p.d; c += 128) {
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
    if (lane16 == 0) p.dsoftmax_sum[(int64_t)bh * p.seqlen_q_rounded + sqi] = acc * p.p_keep;  // p_keep = 1 without dropout
}

// dK/dV kernel geometry.  NW waves x 32 keys per workgroup.  NW = 8 (two waves per SIMD, <= 256
// registers each): K fragments live in registers, V fragments are re-read from the V tile that
// stays in LDS.  NW = 4 (one wave per SIMD, for short key sequences): K and V fragments both in
// registers, the V tile's LDS space is reused for the Q / dO slices.
template <int DP, int NW>
struct BwdCfg {
    static constexpr int NT = NW * 64;
    static constexpr int KB = NW * 32;                        // keys per workgroup
    // (4 waves at d = 128: V in LDS keeps the kernel at 256 registers, so that two workgroups share
    //  a CU and overlap each other's prologues and epilogues -- the variant for short key ranges)
    static constexpr bool V_IN_LDS = NW == 8 || DP >= 128;
    static constexpr int WG_PER_CU = (NW == 4 && DP == 128) ? 2 : 1;
    // d = 256: the dK / dV accumulators of 32 keys x 256 columns are 256 registers, the whole
    // accumulation half of the register file; with S, dP and the K fragments beside them the
    // compiler moves tiles between the register files and scratch all through the step (measured:
    // 111 TFLOP/s backward).  The head dim is therefore cut in DSPLIT column halves, each taken by
    // its own workgroup (blockIdx.z): S and dP are computed in both (12 instead of 8 products per
    // key block and query slice), the accumulators are 128 registers and nothing spills.
    static constexpr int DSPLIT = DP == 256 ? 2 : 1;
    static constexpr int CPR = DP / 8;
    static constexpr int KV_BYTES = KB * DP * 2;              // one of K / V
    // Sub-blocks per slice, 8-wave kernel: 2 (measured on MI355X: +29 % at d = 64, +4 % at d = 128,
    // where the staging registers had to be shared between the chunks to make room).
    static constexpr int QSUB = (NW == 8 && DP <= 128) ? 2 : 1;
    static constexpr int QM = 32 * QSUB;                      // query rows per step
    static constexpr int QT_BYTES = QM * DP * 2;              // one of Q / dO slice
    static constexpr int QBUF_BYTES = 2 * QT_BYTES + 8 * QM;  // Q, dO, lse[QM], delta[QM]
    // LDS map: [K tile | V tile] while loading; then V_IN_LDS: [Q bufs (over K) ... | V tile],
    // else: [unused K | Q bufs (over V)]
    static constexpr int OFF_K = 0;
    static constexpr int OFF_V = KV_BYTES;
    static constexpr int OFF_Q = V_IN_LDS ? 0 : KV_BYTES;
    static constexpr int QSPACE = 2 * QBUF_BYTES > KV_BYTES ? 2 * QBUF_BYTES : KV_BYTES;
    static constexpr int MAIN_LDS = V_IN_LDS ? (QSPACE > KV_BYTES ? QSPACE + KV_BYTES : 2 * KV_BYTES)
                                             : KV_BYTES + QSPACE;
    static constexpr int OFF_VT = V_IN_LDS ? (QSPACE > KV_BYTES ? QSPACE : KV_BYTES) : KV_BYTES;
    static constexpr int OUT_ROWB = DP * 2 + 16;
    static constexpr int OUT_LDS = NW * 32 * OUT_ROWB;
    static constexpr int LDS_BYTES = MAIN_LDS > OUT_LDS ? MAIN_LDS : OUT_LDS;
    static constexpr int KV_CH = KB * CPR / NT;               // chunks per thread per K (or V) tile
    static constexpr int QCH = (QM * CPR + NT - 1) / NT;      // chunks per thread per Q (or dO) slice
    static constexpr int PP_KLDS = 4;                         // two-phase schedule: K fragments per wave kept in LDS
    static constexpr int PP_LDS_BYTES = LDS_BYTES + (NW / 2) * PP_KLDS * 1024;   // (waves 0-3 use the gap behind the ring)
};

template <int DP, bool BF16, int NW, bool FAST, bool DROP, bool HS, bool PP = false, int DV = DP>
__global__ void __launch_bounds__(NW * 64, (BwdCfg<DP, NW>::WG_PER_CU))
bwd_dkdv_kernel(const KParams p, const int n_kblocks) {
    using Cfg = BwdCfg<DP, NW>;
    constexpr int CPR = Cfg::CPR, NT = Cfg::NT, KB = Cfg::KB;
    // DV <= DP: the head dims this instantiation serves, rounded up to 32.  Tiles and fragments keep
    // the layout of DP columns (zero beyond the head dim); the products stop at column DV.
    static_assert(DV % 32 == 0 && DV <= DP && 2 * DV > DP, "DV");
    constexpr int KS = DV / 16, NDB = DV / 32 / Cfg::DSPLIT;   // NDB: 32-column blocks of dK / dV held by this workgroup
    const int db0 = Cfg::DSPLIT > 1 ? (int)blockIdx.z * NDB : 0;   // its first block
    constexpr int QCH = Cfg::QCH, BW_QM = Cfg::QM, QSUB = Cfg::QSUB;
    constexpr bool V_IN_LDS = Cfg::V_IN_LDS;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    BLK_STAMP(0);
    // The kernel arguments are read piece by piece along the prologue, each piece a scalar load
    // with a wait behind it: one word of every 64-byte line is requested here, so that the later
    // loads find their lines in the scalar cache.
    int ka_touch = 0;
    {
        const int __attribute__((address_space(4))) *ka =
            (const int __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
#pragma unroll
        for (int i = 0; i < (int)((sizeof(KParams) + 63) / 64); i++) ka_touch |= ka[16 * i];
    }

    // block -> (batch, kv head, key block).  Blocks b and b+8 share an XCD (round-robin dispatch):
    // give each XCD whole (batch, kv head) pairs, so the key blocks of a pair run side by side on
    // one XCD and the Q / dO slices they all sweep (in the same order, see below) are L2 hits for
    // every block but the first.
    int n_block, hk, bidb;
    {
        const int P = p.b * p.h_k;
        const int L = blockIdx.x;
        const int full = (P / 8) * 8 * n_kblocks;
        int pair;
        if (L < full) {
            // Within an XCD the pairs are taken in groups of G, and within a group the key blocks
            // go heaviest first ACROSS its pairs (key block 0 of every pair, then key block 1, ...):
            // under a causal mask key block n sees (n_kblocks - n) / n_kblocks of the rows, and
            // pair-by-pair order would start the heavy blocks of the next pair only as the light
            // ones of this pair finish (25 instead of 17 block-units for 4 pairs on 32 CUs).  G is
            // kept at about two rounds of the XCD's CUs, so the co-resident blocks still come from
            // few pairs and share their Q / dO slices through L2.
            const int x = L & 7, sl = L >> 3;
            const int ppx = P / 8;
            const int G = max(1, min(ppx, (64 + n_kblocks - 1) / n_kblocks));
            const int g = sl / (G * n_kblocks);
            const int within = sl - g * G * n_kblocks;
            const int Gl = min(G, ppx - g * G);
            n_block = within / Gl;
            pair = (g * G + within % Gl) * 8 + x;
        } else {
            // the last P % 8 pairs, spread over all XCDs: heaviest key blocks first across them
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
    BLK_STAMP(4);
    // dropout: the forward's Philox state; 1 KiB of LDS per wave behind the kernel's own map
    unsigned long long ph_seed = 0, ph_off = 0;
    unsigned *xch = nullptr;
    if constexpr (DROP) {
        ph_seed = p.rng_read ? p.rng_state[0] : p.philox_seed;
        ph_off = p.rng_read ? p.rng_state[1] : p.philox_offset;
        xch = (unsigned *)(smem + Cfg::LDS_BYTES) + 256 * wave;
    }

    // query-slice range that can see this key block
    int m_min = 0, m_max = (sq + BW_QM - 1) / BW_QM;
    if (p.is_causal || p.is_local) m_min = max(0, (k0 + sq - sk - p.wr) / BW_QM);
    if (p.is_local) m_max = min(m_max, (k0 + KB + sq - sk + p.wl + BW_QM - 1) / BW_QM);

    const unsigned short *kg = (const unsigned short *)p.k + bi.k_off(p.k_bs, p.k_rs, bidb) + (int64_t)hk * p.k_hs;
    const unsigned short *vg = (const unsigned short *)p.v + bi.k_off(p.v_bs, p.v_rs, bidb) + (int64_t)hk * p.v_hs;

    // ---- K (and V where it stays in registers): the B-operand fragments of this wave's 32 keys,
    // lane (r, hh) holds K[kw0 + r][16 s + 8 hh ..], straight from global memory; rows past sk and
    // the head-dim padding are zeros.  The V tile (V_IN_LDS) is requested here and written to LDS
    // after the first Q / dO slice has been requested, so that the prologue waits for one memory
    // round trip, not two.
    // (FAST: buffer loads whose descriptor ends at the last valid key row -- no predication, so
    //  all requests go out back to back; a predicated load is a branch, and a copy behind one of
    //  them made the compiler wait for the memory in the middle of the sequence)
    i32x4 kf[KS], vf[V_IN_LDS ? 1 : KS];
    i32x4 vstage[V_IN_LDS ? Cfg::KV_CH : 1];
    // Below d = 128 the tiles go through LDS first and the fragments are read from there: the
    // single round trip measured 7 % slower at d = 64 (1.18 -> 1.27 ms at b4 s4096 h32; the step
    // loop comes out of the compiler differently), against 3 % faster at d = 128.
    constexpr bool VIA_LDS = DP < 128;
    if constexpr (VIA_LDS) {
        i32x4 kk[Cfg::KV_CH], vv[Cfg::KV_CH];
        if constexpr (FAST) {   // unpredicated: all requests go out before the first result is used
            const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void *)kg, 0, (unsigned)(((int64_t)(sk - 1) * p.k_rs + p.d) * 2), 0x00020000);
            const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void *)vg, 0, (unsigned)(((int64_t)(sk - 1) * p.v_rs + p.d) * 2), 0x00020000);
#pragma unroll
            for (int i = 0; i < Cfg::KV_CH; i++) {
                const int c = tid + i * NT;
                const int row = c / CPR, ch = c % CPR;
                const bool ok = ch * 8 < p.d;
                kk[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    k_rsrc, ok ? ((unsigned)(k0 + row) * (unsigned)p.k_rs + (unsigned)ch * 8) * 2 : 0x80000000u, 0, 0));
                vv[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    v_rsrc, ok ? ((unsigned)(k0 + row) * (unsigned)p.v_rs + (unsigned)ch * 8) * 2 : 0x80000000u, 0, 0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < Cfg::KV_CH; i++) {
                const int c = tid + i * NT;
                const int row = c / CPR, ch = c % CPR;
                const bool ok = (k0 + row) < sk && ch * 8 < p.d;
                kk[i] = ok ? *(const i32x4 *)(kg + (int64_t)(k0 + row) * p.k_rs + ch * 8) : i32x4{0, 0, 0, 0};
                vv[i] = ok ? *(const i32x4 *)(vg + (int64_t)(k0 + row) * p.v_rs + ch * 8) : i32x4{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int i = 0; i < Cfg::KV_CH; i++) {
            const int c = tid + i * NT;
            const int off = lds_off<DP>(c / CPR, c % CPR);
            *(i32x4 *)(smem + Cfg::OFF_K + off) = kk[i];
            *(i32x4 *)(smem + Cfg::OFF_VT + off) = vv[i];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KS; s++) {
            kf[s] = *(const i32x4 *)(smem + Cfg::OFF_K + lds_off<DP>(32 * wave + r, 2 * s + hh));
            if constexpr (!V_IN_LDS)
                vf[s] = *(const i32x4 *)(smem + Cfg::OFF_VT + lds_off<DP>(32 * wave + r, 2 * s + hh));
        }
    } else if constexpr (FAST) {
        const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)kg, 0, (unsigned)(((int64_t)(sk - 1) * p.k_rs + p.d) * 2), 0x00020000);
        const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)vg, 0, (unsigned)(((int64_t)(sk - 1) * p.v_rs + p.d) * 2), 0x00020000);
        const unsigned krow = (unsigned)(kw0 + r) * (unsigned)p.k_rs * 2, vrow = (unsigned)(kw0 + r) * (unsigned)p.v_rs * 2;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int col = (2 * s + hh) * 8;
            const bool ok = col < p.d;
            kf[s] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                k_rsrc, ok ? krow + (unsigned)col * 2 : 0x80000000u, 0, 0));
            if constexpr (!V_IN_LDS)
                vf[s] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    v_rsrc, ok ? vrow + (unsigned)col * 2 : 0x80000000u, 0, 0));
        }
        if constexpr (V_IN_LDS) {
#pragma unroll
            for (int i = 0; i < Cfg::KV_CH; i++) {
                const int c = tid + i * NT;
                const int row = c / CPR, ch = c % CPR;
                vstage[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    v_rsrc, ch * 8 < p.d ? ((unsigned)(k0 + row) * (unsigned)p.v_rs + (unsigned)ch * 8) * 2 : 0x80000000u, 0, 0));
            }
        }
    } else {
        const bool rok = kw0 + r < sk;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int col = (2 * s + hh) * 8;
            const bool ok = rok && col < p.d;
            kf[s] = ok ? *(const i32x4 *)(kg + (int64_t)(kw0 + r) * p.k_rs + col) : i32x4{0, 0, 0, 0};
            if constexpr (!V_IN_LDS)
                vf[s] = ok ? *(const i32x4 *)(vg + (int64_t)(kw0 + r) * p.v_rs + col) : i32x4{0, 0, 0, 0};
        }
        if constexpr (V_IN_LDS) {
#pragma unroll
            for (int i = 0; i < Cfg::KV_CH; i++) {
                const int c = tid + i * NT;
                const int row = c / CPR, ch = c % CPR;
                const bool ok = (k0 + row) < sk && ch * 8 < p.d;
                vstage[i] = ok ? *(const i32x4 *)(vg + (int64_t)(k0 + row) * p.v_rs + ch * 8) : i32x4{0, 0, 0, 0};
            }
        }
    }
    auto store_v_tile = [&]() __attribute__((always_inline)) {
        if constexpr (V_IN_LDS && !VIA_LDS) {
#pragma unroll
            for (int i = 0; i < Cfg::KV_CH; i++) {
                const int c = tid + i * NT;
                *(i32x4 *)(smem + Cfg::OFF_VT + lds_off<DP>(c / CPR, c % CPR)) = vstage[i];
            }
        }
    };
    const char *vt = smem + Cfg::OFF_VT;

    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int i = 0; i < NDB; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) { dk[i][j] = 0.f; dv[i][j] = 0.f; }

    const float c2 = p.scale_log2;
    const bool has_alibi = p.alibi_slopes != nullptr;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
    const int n_steps_per_head = max(m_max - m_min, 0);
    // Head split (HS: p.hsplit > 1, blockIdx.y): this workgroup sweeps only heads [h_lo, h_lo + nh)
    // of the GQA group and leaves fp32 partial dK / dV in its slab of p.dk_part / p.dv_part for
    // bwd_kv_reduce_kernel -- for launches with too few (batch, kv head, key block) items to fill
    // the chip (small batch with MQA / GQA).  A template parameter: as a run-time branch its
    // scalars cost the unsplit kernel 5 more spilled SGPRs and 8 % of its time.
    const int gper = HS ? (p.h_ratio + p.hsplit - 1) / p.hsplit : p.h_ratio;
    const int h_lo = HS ? (int)blockIdx.y * gper : 0;
    const int nh = HS ? max(0, min(p.h_ratio, h_lo + gper) - h_lo) : p.h_ratio;
    const int total_steps = n_steps_per_head * nh;


    if constexpr (PP) {
    // ---- PP: the two waves of a SIMD (waves w and w + 4) alternate between a MATRIX phase and a
    // VECTOR phase, in anti-phase, with a workgroup barrier after every phase.  Sub-blocks of 32
    // query rows; for sub-block u a wave runs
    //   V(u): P = exp2(S), dS = P (dP - delta), both rounded to 16 bit   (vector work only)
    //   M(u): dV += dO^T P, dK += Q^T dS of sub-block u, then S and dP of sub-block u + 1
    //         (32 MFMAs, LDS reads, no dependence on this phase's vector results)
    // and waves 4-7 run one phase behind waves 0-3, so that at any time one wave of a SIMD feeds
    // the matrix pipe while its partner's exponentials use the vector issue slots the MFMAs leave.
    // The 32-row Q / dO slices (+ their LSE and delta rows) travel through a ring of four LDS
    // buffers, requested three sub-blocks ahead by loads that write LDS directly; every wave waits
    // for its own pieces (counted vmcnt) in front of the barrier that precedes their first read.
    static_assert(FAST && DP == 128 && NW == 8 && !DROP, "two-phase dK/dV schedule: d = 128, 8 waves");
    store_v_tile();
    __syncthreads();
    // Ring layout: the four 256-byte row-constant blocks (lse[32], delta[32] as stored in global
    // memory) first, then the four [Q rows | dO rows] buffers -- every offset the reads add to
    // their per-lane address then fits the 16-bit immediate field of the DS instructions.
    constexpr int SL_DO = 32 * DP * 2;        // byte offset of the dO rows in a ring buffer
    constexpr int SL_BYTES = 2 * SL_DO;
    constexpr int NRING = 3;
    constexpr int RING0 = 4 * 256;
    constexpr int KL0 = RING0 + NRING * SL_BYTES;   // K fragments of waves 0-3; those of waves 4-7 sit behind the V tile
    static_assert(KL0 + 4 * Cfg::PP_KLDS * 1024 <= Cfg::OFF_VT, "ring and K fragments sit in front of the V tile");
    static_assert(Cfg::PP_LDS_BYTES <= 160 * 1024, "LDS");
    const bool grp_b = wave >= 4;
    int m_lo = 0, m_hi = (sq + 31) / 32;
    if (p.is_causal || p.is_local) m_lo = max(0, (k0 + sq - sk - p.wr) / 32);
    if (p.is_local) m_hi = min(m_hi, (k0 + KB + sq - sk + p.wl + 31) / 32);
    const int T = max(m_hi - m_lo, 0) * nh;
    const float nis = -1.f / p.scale;
    auto advance = [&](int &hh_, int &m_) __attribute__((always_inline)) {
        if (++hh_ == nh) {
            hh_ = 0;
            --m_;
        }
    };
    // The last KLDS of a wave's 8 K fragments are kept in LDS (lane order, 1 KiB each, behind the
    // V tile) and re-read per sub-block like the V fragments: the registers they would take are
    // what lets the LDS reads run ahead of the MFMAs without spilling.
    constexpr int KLDS = Cfg::PP_KLDS;
    const int kl_base = (wave < 4 ? KL0 + wave * KLDS * 1024 : Cfg::LDS_BYTES + (wave - 4) * KLDS * 1024) + lane * 16;
    __syncthreads();  // every wave has read its K fragments from the K tile, which the stores below overwrite
#pragma unroll
    for (int i = 0; i < KLDS; i++) *(i32x4 *)(smem + kl_base + i * 1024) = kf[KS - KLDS + i];
    __syncthreads();  // every wave has its K fragments: the K tile's space becomes the ring
    if (T > 0) {
    // Waves 4-7 do all the loading (at the top of their vector phase, where they have scalar and
    // memory issue slots to spare): each fetches two 1-KiB pieces (4 rows) of the Q rows and two of
    // the dO rows of a slice, wave 4 also the 32 LSE and 32 delta values.  Per-lane part of the
    // offsets: lane -> (row, chunk) of its LDS slot.
    const int pc0 = wave & 3;                             // this wave's pieces: pc0 and pc0 + 4 (rows + 16)
    const int row0 = 4 * pc0 + (lane >> 4);
    const int chs = (lane & 15) ^ swz<DP>(row0);          // (the swizzle of row + 16 is the same)
    const bool colok = chs * 8 < p.d;
    const unsigned q_vo = colok ? (unsigned)((row0 * (int)p.q_rs + chs * 8) * 2) : 0x80000000u;
    const unsigned do_vo = colok ? (unsigned)((row0 * (int)p.do_rs + chs * 8) * 2) : 0x80000000u;
    int is_h = 0, is_m = m_hi - 1, is_t = 0;  // next sub-block to request
    // Byte addresses of the batch entry's Q, dO, LSE and delta; a head's descriptor is that base plus
    // a 32-bit head offset (the dispatch checks that it fits), ending at the head's last valid row, so
    // that rows past sq read as zeros.
    const int bidh0 = hk * p.h_ratio + h_lo;
    const char *q_b = (const char *)((const unsigned short *)p.q + bi.q_off(p.q_bs, p.q_rs, bidb));
    const char *do_b = (const char *)((const unsigned short *)p.dout + bi.q_off(p.do_bs, p.do_rs, bidb));
    const char *l_b = (const char *)(p.lse + (int64_t)bidb * p.h * p.seqlen_q);
    const char *d_b = (const char *)(p.dsoftmax_sum + (int64_t)bidb * p.h * p.seqlen_q_rounded);
    const unsigned q_bytes = (unsigned)(((int64_t)(sq - 1) * p.q_rs + p.d) * 2);
    const unsigned do_bytes = (unsigned)(((int64_t)(sq - 1) * p.do_rs + p.d) * 2);
    const unsigned q_hs2 = (unsigned)p.q_hs * 2, do_hs2 = (unsigned)p.do_hs * 2;
    i32x4 rq_r, rdo_r, rl_r, rd_r;   // descriptors of the head being requested
    auto set_head = [&](const int gh) __attribute__((always_inline)) {
        const unsigned bidh = (unsigned)(bidh0 + gh);
        rq_r = make_rsrc(q_b + (size_t)(bidh * q_hs2), q_bytes);
        rdo_r = make_rsrc(do_b + (size_t)(bidh * do_hs2), do_bytes);
        rl_r = make_rsrc(l_b + (size_t)(bidh * (unsigned)p.seqlen_q * 4), (unsigned)sq * 4);
        rd_r = make_rsrc(d_b + (size_t)(bidh * (unsigned)p.seqlen_q_rounded * 4), (unsigned)sq * 4);
    };
    set_head(0);
    // Requests past the last sub-block are issued all the same, out of range (they write zeros
    // into a buffer nobody reads), so that the count of outstanding loads stays the same.
    auto issue = [&](auto RINGC) __attribute__((always_inline)) {
        constexpr unsigned RB = RING0 + decltype(RINGC)::value * SL_BYTES, RC = decltype(RINGC)::value * 256;
        const bool valid = is_t < T;
        if (nh > 1 && valid) set_head(is_h);
        const unsigned rows = (unsigned)((valid ? is_m : 0) * 32);
        if (wave == 4) {
            const unsigned rv = valid ? (rows + (unsigned)(lane & 31)) * 4 : 0x80000000u;
            // lane L writes dword L of the 256 bytes
            if (lane < 32) dma_4b_to_lds(rl_r, rv, RC);
            else dma_4b_to_lds(rd_r, rv, RC);
        }
        const unsigned qv = valid ? q_vo + rows * (unsigned)p.q_rs * 2 : 0x80000000u;
        const unsigned dv_ = valid ? do_vo + rows * (unsigned)p.do_rs * 2 : 0x80000000u;
        const unsigned lb = RB + (unsigned)pc0 * 1024u;
        dma_16b_x4_to_lds(rq_r, qv, valid ? qv + 16 * (unsigned)p.q_rs * 2 : 0x80000000u, lb, lb + 4096u,
                          rdo_r, dv_, valid ? dv_ + 16 * (unsigned)p.do_rs * 2 : 0x80000000u, lb + SL_DO, lb + SL_DO + 4096u);
        if (valid) advance(is_h, is_m);
        ++is_t;
    };
    // Once a slice has landed, wave 4 turns its row constants into what the accumulators start
    // from: -LSE / scale (lanes 0-31) and -delta (lanes 32-63), in place.
    auto finish_rowc = [&](auto RINGC) __attribute__((always_inline)) {
        constexpr unsigned RC = decltype(RINGC)::value * 256;
        if (wave == 4) {
            float *pr = (float *)(smem + RC) + lane;
            const float x = *pr;
            *pr = lane < 32 ? x * nis : -x;
        }
    };
    // wait until at most K of this wave's requests (4 loads each, 6 on wave 4) are in flight
    auto wait_keep = [&](auto KC) __attribute__((always_inline)) {
        constexpr int K = decltype(KC)::value;
        if (wave == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * K) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * K) : "memory");
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    f32x16 s, dp;            // S and dP of the sub-block whose V phase comes next
    i32x4 pf[2], dsf[2];     // P and dS of the sub-block whose dV / dK products come next
    bool skip_cur = true, skip_nx = true;
    int cur_h = 0, cur_m = m_hi - 1, nx_h = 0, nx_m = m_hi - 1;
    float alibi_cur = 0.f, alibi_nx = 0.f;

    // LDS addresses are rebuilt per phase from four registers each instead of being kept (the
    // compiler would hoist all 24 of them out of the loop, and the registers are what limits how far
    // ahead of the MFMAs the reads can run).  Row read of k-step ks: row r, chunk (2 ks + hh) ^
    // swz(r): bit b of ks flips chunk bit b + 1, i.e. moves the address by +-(32 << b) bytes with a
    // sign that is a constant of the lane.  Transposed read of d block db, row pair e: see the
    // tr_* comments below.  `opaque` keeps the compiler from precomputing the sums.
    const int xr = swz<DP>(r) ^ hh;
    const int ra0 = r * (DP * 2) + 16 * xr;
    const int rd0 = (xr & 2) ? -32 : 32, rd1 = (xr & 4) ? -64 : 64, rd2 = (xr & 8) ? -128 : 128;
    auto row_addr = [&](const int base, const int ks) __attribute__((always_inline)) {
        return base + ((ks & 1) ? rd0 : 0) + ((ks & 2) ? rd1 : 0) + ((ks & 4) ? rd2 : 0);
    };
    // transposed read (d block db, rows 4 hh + tr_q + 8 e + 16 ks2, chunk 4 db + 2 tr_g + (tr_p >> 1)):
    //   offset = row * 256 + 64 (db ^ tr_q) + 16 (low ^ ((hh + 2 e) & 3)) + 8 (tr_p & 1),  low = 2 tr_g + (tr_p >> 1)
    const int tlow = (2 * tr_g + (tr_p >> 1)) ^ hh;
    const int ta0 = (4 * hh + tr_q) * (DP * 2) + 64 * tr_q + 16 * tlow + 8 * (tr_p & 1);
    const int te0 = (tr_q & 1) ? -64 : 64, te1 = (tr_q & 2) ? -128 : 128;
    const int tf = ((tlow & 2) ? -32 : 32) + 8 * (DP * 2);   // e = 1: rows + 8, chunk bit 1 flipped
    auto opaque = [](int x) __attribute__((always_inline)) { asm volatile("" : "+v"(x)); return x; };
    const int v_base = Cfg::OFF_VT + 32 * wave * (DP * 2);

    // S and dP of sub-block (nx_h, nx_m), read from ring buffer RING
    auto m_part2 = [&](auto RINGC) __attribute__((always_inline)) {
        constexpr int RB = RING0 + decltype(RINGC)::value * SL_BYTES, RC = decltype(RINGC)::value * 256;
        const int q0 = nx_m * 32;
        skip_nx = false;
        if (p.is_causal || p.is_local) skip_nx = kw0 >= q0 + 32 + sk - sq + p.wr;
        if (p.is_local) skip_nx = skip_nx || (kw0 + 32 <= q0 + sk - sq - p.wl);
        if (has_alibi) alibi_nx = p.alibi_slopes[(int64_t)bidb * p.alibi_bs + hk * p.h_ratio + h_lo + nx_h] / p.scale;
        if (skip_nx) return;
        const float *lse_s = (const float *)(smem + RC), *dl_s = lse_s + 32;
        const int ra = opaque(ra0);
#ifndef VFA_PP_PFQ
#define VFA_PP_PFQ 2
#define VFA_PP_PFD 2
#endif
        constexpr int PFQ = VFA_PP_PFQ, PFD = VFA_PP_PFD;   // k-steps the reads run ahead of the MFMAs (S chain, dP chain)
        i32x4 qa[KS], da[KS], vb[KS], kl[KS];
        auto rd_q = [&](const int ks) __attribute__((always_inline)) {
            qa[ks] = *(const i32x4 *)(smem + RB + row_addr(ra, ks));
            if (ks >= KS - KLDS) kl[ks] = *(const i32x4 *)(smem + kl_base + (ks - (KS - KLDS)) * 1024);
        };
#pragma unroll
        for (int ks = 0; ks < PFQ; ks++) rd_q(ks);
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const f32x4 ls = *(const f32x4 *)(lse_s + 8 * g4 + 4 * hh);
            const f32x4 dl = *(const f32x4 *)(dl_s + 8 * g4 + 4 * hh);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                s[4 * g4 + t] = ls[t];     // (stored as -LSE / scale and -delta: finish_rowc)
                dp[4 * g4 + t] = dl[t];
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if (ks + PFQ < KS) rd_q(ks + PFQ);
            if (ks + PFD >= KS) {  // the dP chain's first fragments behind the last Q fragments
                const int k2 = ks + PFD - KS;
                da[k2] = *(const i32x4 *)(smem + RB + SL_DO + row_addr(ra, k2));
                vb[k2] = *(const i32x4 *)(smem + v_base + row_addr(ra, k2));
            }
            s = mma32<BF16>(qa[ks], ks >= KS - KLDS ? kl[ks] : kf[ks], s);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if (ks + PFD < KS) {
                da[ks + PFD] = *(const i32x4 *)(smem + RB + SL_DO + row_addr(ra, ks + PFD));
                vb[ks + PFD] = *(const i32x4 *)(smem + v_base + row_addr(ra, ks + PFD));
            }
            dp = mma32<BF16>(da[ks], vb[ks], dp);
        }
        // the same order for the scheduler: first fragments and the 8 row-constant reads, then one
        // MFMA per k-step with the reads of the k-step PFQ / PFD ahead
        constexpr int NKL0 = PFQ > KS - KLDS ? PFQ - (KS - KLDS) : 0;   // K fragments among the first PFQ k-steps
        sched_group<0x100, 8 + PFQ + NKL0>();
        static_for<0, KS>([&](auto KC) {
            constexpr int ks = decltype(KC)::value;
            constexpr int n = (ks + PFQ < KS ? 1 + (ks + PFQ >= KS - KLDS ? 1 : 0) : 0) + (ks + PFD >= KS ? 2 : 0);
            sched_group<0x100, n>();
            sched_group<0x008, 1>();
        });
        static_for<0, KS>([&](auto KC) {
            constexpr int ks = decltype(KC)::value;
            sched_group<0x100, (ks + PFD < KS ? 2 : 0)>();
            sched_group<0x008, 1>();
        });
    };
    // dV and dK products of sub-block (cur_h, cur_m), transposed reads of ring buffer RING
    auto m_part1 = [&](auto RINGC) __attribute__((always_inline)) {
        constexpr int RB = RING0 + decltype(RINGC)::value * SL_BYTES;
        if (skip_cur) return;
        const int ta = opaque(ta0);
        constexpr int NG = 2 * NDB, PF = 2;   // groups (d block, k-step) of 2 MFMAs; read-ahead in groups
        i32x4 dot[NG], qt[NG];
        auto rd = [&](const int g) __attribute__((always_inline)) {
            const int db = g >> 1, ks2 = g & 1;
            const int o0 = ta + ((db & 1) ? te0 : 0) + ((db & 2) ? te1 : 0) + RB + 16 * ks2 * (DP * 2);
            const int o1 = o0 + tf;
            const i32x2 a0 = __builtin_bit_cast(i32x2, lds_read_tr16(smem + SL_DO + o0));
            const i32x2 a1 = __builtin_bit_cast(i32x2, lds_read_tr16(smem + SL_DO + o1));
            const i32x2 b0 = __builtin_bit_cast(i32x2, lds_read_tr16(smem + o0));
            const i32x2 b1 = __builtin_bit_cast(i32x2, lds_read_tr16(smem + o1));
            dot[g] = i32x4{a0[0], a0[1], a1[0], a1[1]};
            qt[g] = i32x4{b0[0], b0[1], b1[0], b1[1]};
        };
#pragma unroll
        for (int g = 0; g < PF; g++) rd(g);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            if (g + PF < NG) rd(g + PF);
            dv[g >> 1] = mma32<BF16>(dot[g], pf[g & 1], dv[g >> 1]);
            dk[g >> 1] = mma32<BF16>(qt[g], dsf[g & 1], dk[g >> 1]);
        }
        // the order above, imposed on the scheduler (it would otherwise sink every read to just in
        // front of its MFMA and leave the whole LDS latency exposed): reads PF groups ahead
        sched_group<0x100, 4 * PF>();
        static_for<0, NG>([&](auto GC) {
            constexpr int g = decltype(GC)::value;
            if constexpr (g + PF < NG) {
                sched_group<0x100, 2>(); sched_group<0x008, 1>();
                sched_group<0x100, 2>(); sched_group<0x008, 1>();
            } else {
                sched_group<0x008, 2>();
            }
        });
    };
    // P and dS of sub-block (cur_h, cur_m) from its S and dP (forward's mask and bias)
    auto v_phase = [&]() __attribute__((always_inline)) {
        if (skip_cur) return;
        const int q0 = cur_m * 32;
        const int col = kw0 + r;
        const bool masked_tile = has_alibi || (kw0 + 32 > sk) || (q0 + 32 > sq) ||
                                 ((p.is_causal || p.is_local) && (kw0 + 32 > q0 + 1 + sk - sq + p.wr)) ||
                                 (p.is_local && (kw0 < q0 + 31 + sk - sq - p.wl));
        if (masked_tile) {  // (the arithmetic of the step loop below)
            constexpr int BIG = 1 << 28;
            const int shift = q0 + 4 * hh + sk - sq;
            int lo_r = (p.is_causal || p.is_local) ? col - shift - p.wr : -BIG;
            int hi_r = p.is_local ? col - shift + p.wl : BIG;
            hi_r = min(hi_r, sq - q0 - 4 * hh - 1);
            lo_r = max(lo_r, -BIG);
            if (col >= sk || hi_r < lo_r) { lo_r = BIG; hi_r = BIG; }
            const unsigned span = (unsigned)(hi_r - lo_r);
            if (has_alibi) {
                if (p.is_causal) {
                    const float ab = alibi_cur * (float)col;
#pragma unroll
                    for (int j = 0; j < 16; j++) s[j] += ab;
                } else {
                    const int dc = shift - col;
#pragma unroll
                    for (int j = 0; j < 16; j++) s[j] -= alibi_cur * fabsf((float)((j & 3) + 8 * (j >> 2) + dc));
                }
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int roff = (j & 3) + 8 * (j >> 2);
                s[j] = ((unsigned)(roff - lo_r) <= span) ? s[j] : -INFINITY;
            }
        }
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int j = 8 * ks2 + 2 * w;
                const float p0 = __builtin_amdgcn_exp2f(s[j] * c2), p1 = __builtin_amdgcn_exp2f(s[j + 1] * c2);
                pf[ks2][w] = pack2<BF16>(p0, p1);
                dsf[ks2][w] = pack2<BF16>(p0 * dp[j], p1 * dp[j + 1]);
            }
    };

#ifdef VFA_STAMP
    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
    if (grp_b) {
        issue(I0{}); issue(I1{});
        wait_keep(I1{});
        finish_rowc(I0{});
    }
    phase_barrier();                 // slice 0 is in LDS
    if (grp_b) phase_barrier();      // waves 4-7 run one phase behind
    m_part2(I0{});
    if (grp_b) {
        wait_keep(I0{});             // slice 1
        finish_rowc(I1{});
    }
    phase_barrier();
    skip_cur = skip_nx; alibi_cur = alibi_nx;
    advance(nx_h, nx_m);
    // One sub-block: u is its index, RING = u % 3 the buffer it sits in.  Slice u + 2 goes into
    // the buffer of slice u - 1, whose last reader (a wave of 4-7 in M(u - 1)) is done when the
    // barrier in front of this point has been passed; it is waited for a phase pair later, at the
    // end of M(u), one barrier ahead of its first reader (a wave of 0-3 in M(u + 1)).
    auto sub_block = [&](const int u, auto RINGC) __attribute__((always_inline)) {
        constexpr int RING = decltype(RINGC)::value;
        using RN = std::integral_constant<int, (RING + 1) % NRING>;   // sub-block u + 1
        using RI = std::integral_constant<int, (RING + 2) % NRING>;   // sub-block u + 2 (requested now)
        STAMP(7);
        if (grp_b) issue(RI{});
        STAMP(0);
        v_phase();
        STAMP(1);
        phase_barrier();
        STAMP(2);
        __builtin_amdgcn_s_setprio(VFA_PP_PRIO);   // the matrix phase wins the issue arbitration against the partner's vector work
        m_part1(RINGC);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3);
        if (u + 1 < T) m_part2(RN{});
        STAMP(4);
        __builtin_amdgcn_s_setprio(0);
        if (grp_b) {
            wait_keep(I0{});
            finish_rowc(RI{});
        }
        phase_barrier();
        STAMP(5);
        cur_h = nx_h; cur_m = nx_m; skip_cur = skip_nx; alibi_cur = alibi_nx;
        advance(nx_h, nx_m);
    };
    for (int u = 0;;) {
        sub_block(u, I0{}); if (++u >= T) break;
        sub_block(u, I1{}); if (++u >= T) break;
        sub_block(u, I2{}); if (++u >= T) break;
    }
#ifdef VFA_STAMP
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long *dst = g_stamps_dkdv + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 7; i++) dst[i] = acc_[i];
        dst[7] = (unsigned long long)T;
    }
#endif
    if (!grp_b) phase_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the out-of-range requests still write (zeros) into the ring
    }  // T > 0
    }  // PP
    if constexpr (!PP) {
    // ---- Q / dO slice staging (register prefetch, two LDS buffers) ----
    float rowc = 0.f;  // lse (tid < QM) or delta (QM <= tid < 2 QM) of the slice
    // From d = 128 the slice carries -LSE / scale and the S accumulator starts from it, so that
    // P = exp2(S * c) needs no second operand (measured: +2.5 % at d = 128; at d = 64, where the
    // short S chain then waits on that LDS read, -13 %).
    constexpr bool LSE_IN_ACC = DP >= 128;
    // Step -> (32-row slice, query head of the GQA group).  Slices are swept from the last one
    // down, and all heads of the group take a slice before the next slice: under a causal mask
    // every key block sees the last rows, so the co-resident blocks of a pair read the same
    // (slice, head) at about the same time and share it through L2 -- head-major order lets the
    // shorter sweeps of the later key blocks run ahead onto other heads (measured with 4 heads per
    // group, b4 s4096 d128 causal: see the dK/dV figures in the design notes).  Kept as counters (an
    // integer division per step costs more instructions than the step's address arithmetic).
    auto advance = [&](int &hh_, int &m_) {
        if (++hh_ == nh) {
            hh_ = 0;
            --m_;
        }
    };
    // FAST: Q / dO rows come through buffer loads whose descriptor (rebuilt per head, it is four
    // scalar words) ends at the last valid query row, so the row tail and the head-dim padding
    // read as zeros with no predication and no 64-bit address arithmetic per chunk.
    constexpr bool HOIST_VOFF = DP <= 64;
    auto voff_of = [&](int i, int64_t rs) -> unsigned {
        const int c = tid + i * NT;
        const int row = c / CPR, ch = c % CPR;
        const bool ok = c < BW_QM * CPR && ch * 8 < p.d;
        return ok ? (unsigned)((row * (int)rs + ch * 8) * 2) : 0x80000000u;
    };
    unsigned q_voff[HOIST_VOFF ? QCH : 1], do_voff[HOIST_VOFF ? QCH : 1];
    if constexpr (HOIST_VOFF) {
#pragma unroll
        for (int i = 0; i < QCH; i++) {
            q_voff[i] = voff_of(i, p.q_rs);
            do_voff[i] = voff_of(i, p.do_rs);
        }
    }
    // SPLIT (two chunks of 32 rows, two sub-blocks): chunk 0 of the next slice is loaded at the top
    // of a step and written to LDS after the first sub-block, chunk 1 is loaded then and written
    // at the end, so one set of staging registers serves both (the registers are what limits the
    // slice height at d = 128).
    // DIRECT (buffer-load path, d = 128): the Q / dO slices go from global memory straight into LDS
    // (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction in lane order, each lane fetching
    // the chunk the XOR swizzle places at its slot): no staging registers and no ds_write.
    // Measured on MI355X: -5..7 % kernel time at d = 128, +8 % at d = 64 (kept on registers).
    constexpr bool DIRECT = FAST && (DP == 128 || DP == 256);
    constexpr bool SPLIT = !DIRECT && QCH == 2 && QSUB == 2;
    constexpr int NSTG = SPLIT ? 1 : QCH;  // staging register sets
    i32x4 qreg[NSTG], doreg[NSTG];
    auto load_chunk = [&](const int gh, const int m, const int i, const int slot) {
        const int bidh = hk * p.h_ratio + h_lo + gh, q0 = m * BW_QM;
        const unsigned short *qg = (const unsigned short *)p.q + bi.q_off(p.q_bs, p.q_rs, bidb) + (int64_t)bidh * p.q_hs;
        const unsigned short *dog = (const unsigned short *)p.dout + bi.q_off(p.do_bs, p.do_rs, bidb) + (int64_t)bidh * p.do_hs;
        if constexpr (FAST) {
            const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void *)qg, 0, (unsigned)(((int64_t)(sq - 1) * p.q_rs + p.d) * 2), 0x00020000);
            const auto do_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void *)dog, 0, (unsigned)(((int64_t)(sq - 1) * p.do_rs + p.d) * 2), 0x00020000);
            const unsigned q_t = (unsigned)q0 * (unsigned)p.q_rs * 2, do_t = (unsigned)q0 * (unsigned)p.do_rs * 2;
            const unsigned qv = HOIST_VOFF ? q_voff[HOIST_VOFF ? i : 0] : voff_of(i, p.q_rs);
            const unsigned dv_ = HOIST_VOFF ? do_voff[HOIST_VOFF ? i : 0] : voff_of(i, p.do_rs);
            qreg[slot] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(q_rsrc, qv + q_t, 0, 0));
            doreg[slot] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(do_rsrc, dv_ + do_t, 0, 0));
        } else {
            const int c = tid + i * NT;
            const int row = c / CPR, ch = c % CPR;
            const bool ok = c < BW_QM * CPR && (q0 + row) < sq && ch * 8 < p.d;
            qreg[slot] = ok ? *(const i32x4 *)(qg + (int64_t)(q0 + row) * p.q_rs + ch * 8) : i32x4{0, 0, 0, 0};
            doreg[slot] = ok ? *(const i32x4 *)(dog + (int64_t)(q0 + row) * p.do_rs + ch * 8) : i32x4{0, 0, 0, 0};
        }
    };
    constexpr int RPI = 64 / CPR;                               // rows per wave instruction
    constexpr int NI = DIRECT ? BW_QM / (RPI * NW) : 1;         // instructions per wave, slice, tensor
    // The rows of a wave's instructions are RPI * NW apart: a multiple of 16 keeps the swizzle of
    // the lane's row (one offset serves all instructions); 8 rows apart (d = 256, 4 waves) it
    // alternates between two values, one for the even and one for the odd instructions.
    constexpr int NVAR = (RPI * NW) % 16 == 0 ? 1 : 2;
    static_assert(!DIRECT || (BW_QM % (RPI * NW) == 0 && (RPI * NW) % 8 == 0), "direct slice loads");
    unsigned dq_voff = 0, ddo_voff = 0, dq_voff1 = 0, ddo_voff1 = 0;   // (the ...1 pair: odd instructions, NVAR = 2)
    if constexpr (DIRECT) {
        const int row0 = RPI * wave + lane / CPR;
        const int ch = (lane % CPR) ^ swz<DP>(row0);
        const bool ok = ch * 8 < p.d;
        dq_voff = ok ? (unsigned)((row0 * (int)p.q_rs + ch * 8) * 2) : 0x80000000u;
        ddo_voff = ok ? (unsigned)((row0 * (int)p.do_rs + ch * 8) * 2) : 0x80000000u;
        if constexpr (NVAR == 2) {
            const int ch1 = (lane % CPR) ^ swz<DP>(row0 + RPI * NW);
            const bool ok1 = ch1 * 8 < p.d;
            dq_voff1 = ok1 ? (unsigned)((row0 * (int)p.q_rs + ch1 * 8) * 2) : 0x80000000u;
            ddo_voff1 = ok1 ? (unsigned)((row0 * (int)p.do_rs + ch1 * 8) * 2) : 0x80000000u;
        }
    }
    auto load_slice_direct = [&](const int gh, const int m, const int buf) __attribute__((always_inline)) {
        if constexpr (DIRECT) {
            typedef __attribute__((address_space(3))) void *lds_ptr;
            const int bidh = hk * p.h_ratio + h_lo + gh, q0 = m * BW_QM;
            const unsigned short *qg = (const unsigned short *)p.q + bi.q_off(p.q_bs, p.q_rs, bidb) + (int64_t)bidh * p.q_hs;
            const unsigned short *dog = (const unsigned short *)p.dout + bi.q_off(p.do_bs, p.do_rs, bidb) + (int64_t)bidh * p.do_hs;
            const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void *)qg, 0, (unsigned)(((int64_t)(sq - 1) * p.q_rs + p.d) * 2), 0x00020000);
            const auto do_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void *)dog, 0, (unsigned)(((int64_t)(sq - 1) * p.do_rs + p.d) * 2), 0x00020000);
            char *qb = smem + Cfg::OFF_Q + buf * Cfg::QBUF_BYTES;
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const unsigned rows = (unsigned)(q0 + RPI * NW * i);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rsrc, (lds_ptr)(qb + (wave + NW * i) * 1024), 16,
                                                         ((NVAR == 2 && (i & 1)) ? dq_voff1 : dq_voff) + rows * (unsigned)p.q_rs * 2, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(do_rsrc, (lds_ptr)(qb + Cfg::QT_BYTES + (wave + NW * i) * 1024), 16,
                                                         ((NVAR == 2 && (i & 1)) ? ddo_voff1 : ddo_voff) + rows * (unsigned)p.do_rs * 2, 0, 0, 0);
            }
        }
    };
    auto load_rowc = [&](const int gh, const int m) {
        const int bidh = hk * p.h_ratio + h_lo + gh, q0 = m * BW_QM;
        if (tid < 2 * BW_QM) {
            const int row = q0 + (tid & (BW_QM - 1));
            const bool ok = row < sq;
            if (tid < BW_QM) {
                const float l = ok ? p.lse[((int64_t)bidb * p.h + bidh) * p.seqlen_q + row] : INFINITY;
                rowc = l;  // (turned into -LSE / scale when it is stored: touching it here would
                           //  make this wave wait for the load it has just issued)
            }
            else rowc = ok ? p.dsoftmax_sum[((int64_t)bidb * p.h + bidh) * p.seqlen_q_rounded + row] : 0.f;
        }
    };
    // whole slice at once (prologue, and every step when !SPLIT)
    auto load_slice = [&](const int gh, const int m) {
        static_assert(!SPLIT || QCH == 2, "split staging is written for two chunks");
        if constexpr (!SPLIT) {
#pragma unroll
            for (int i = 0; i < QCH; i++) load_chunk(gh, m, i, i);
        }
        load_rowc(gh, m);
    };
    // LDS store offsets are loop invariant; at d = 256 the registers to keep them are not there
    constexpr bool HOIST_ST = DP <= 64;
    int st_off[HOIST_ST ? QCH : 1];
    if constexpr (HOIST_ST) {
#pragma unroll
        for (int i = 0; i < QCH; i++) st_off[i] = lds_off<DP>((tid + i * NT) / CPR, (tid + i * NT) % CPR);
    }
    auto store_chunk = [&](int buf, const int i, const int slot) {
        char *qb = smem + Cfg::OFF_Q + buf * Cfg::QBUF_BYTES;
        if (tid + i * NT < BW_QM * CPR) {
            const int off = HOIST_ST ? st_off[HOIST_ST ? i : 0]
                                     : lds_off<DP>((tid + i * NT) / CPR, (tid + i * NT) % CPR);
            *(i32x4 *)(qb + off) = qreg[slot];
            *(i32x4 *)(qb + Cfg::QT_BYTES + off) = doreg[slot];
        }
    };
    auto store_rowc = [&](int buf) {
        char *qb = smem + Cfg::OFF_Q + buf * Cfg::QBUF_BYTES;
        if (tid < 2 * BW_QM)
            // delta is stored negated: the dP accumulator starts from it as it is read
            ((float *)(qb + 2 * Cfg::QT_BYTES))[tid] = tid >= BW_QM ? -rowc : (LSE_IN_ACC ? -rowc / p.scale : rowc);
    };
    auto store_slice = [&](int buf) {
        if constexpr (!SPLIT) {
#pragma unroll
            for (int i = 0; i < QCH; i++) store_chunk(buf, i, i);
        }
        store_rowc(buf);
    };

    int cur_h = 0, cur_m = m_max - 1, nxt_h = 0, nxt_m = m_max - 1;
    advance(nxt_h, nxt_m);
    if constexpr (DIRECT) {
        if (total_steps > 0) {
            load_rowc(cur_h, cur_m);
            load_slice_direct(cur_h, cur_m, 0);
        }
    } else {
        if (total_steps > 0) load_slice(cur_h, cur_m);
    }
    if (SPLIT && total_steps > 0) load_chunk(cur_h, cur_m, 0, 0);
    if constexpr (VIA_LDS) __syncthreads();  // every wave has its fragments: the freed tile space becomes the Q buffers
    asm volatile("" ::"s"(ka_touch));
    BLK_STAMP(5);
    store_v_tile();
    if constexpr (DIRECT) {
        if (total_steps > 0) store_rowc(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    } else if (total_steps > 0) store_slice(0);
    BLK_STAMP(6);
    if (SPLIT && total_steps > 0) {
        store_chunk(0, 0, 0);
        load_chunk(cur_h, cur_m, 1, 0);
        store_chunk(0, 1, 0);
    }
    __syncthreads();

#ifdef VFA_STAMP
    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
    BLK_STAMP(1);
    int buf = 0;
    for (int step = 0; step < total_steps; step++) {
        const int bidh = hk * p.h_ratio + h_lo + cur_h, q0s = cur_m * BW_QM;
        const bool has_next = step + 1 < total_steps;
        STAMP(7);
        if (has_next) {
            if constexpr (DIRECT) {
                load_slice_direct(nxt_h, nxt_m, buf ^ 1);
                load_rowc(nxt_h, nxt_m);
            } else {
                load_slice(nxt_h, nxt_m);
            }
            if constexpr (SPLIT) load_chunk(nxt_h, nxt_m, 0, 0);
        }
        STAMP(0);

        const char *qb0 = smem + Cfg::OFF_Q + buf * Cfg::QBUF_BYTES;
        const float *lse_s0 = (const float *)(qb0 + 2 * Cfg::QT_BYTES);
        float alibi = 0.f;
        if (has_alibi) alibi = p.alibi_slopes[(int64_t)bidb * p.alibi_bs + bidh] / p.scale;

#pragma unroll
        for (int sub = QSUB - 1; sub >= 0; sub--) {  // 32-row sub-blocks, last rows first
        const int q0 = q0s + 32 * sub;
        const char *qb = qb0 + 32 * sub * (DP * 2);
        const char *dob = qb + Cfg::QT_BYTES;
        const float *lse_s = lse_s0 + 32 * sub;
        const float *dl_s = lse_s0 + BW_QM + 32 * sub;

        // a wave whose 32 keys are invisible to all 32 rows of this sub-block only helps with staging
        bool skip = false;
        if (p.is_causal || p.is_local) skip = kw0 >= q0 + 32 + sk - sq + p.wr;
        if (p.is_local) skip = skip || (kw0 + 32 <= q0 + sk - sq - p.wl);
        if (!skip) {
        // ---- S[q][key] = Q K^T (- LSE[q] / scale); key on the lane ----
        f32x16 s;
        if constexpr (LSE_IN_ACC) {
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const f32x4 ls = *(const f32x4 *)(lse_s + 8 * g4 + 4 * hh);
#pragma unroll
                for (int t = 0; t < 4; t++) s[4 * g4 + t] = ls[t];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) s[j] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const i32x4 qa = *(const i32x4 *)(qb + lds_off<DP>(r, 2 * ks + hh));
            s = mma32<BF16>(qa, kf[ks], s);
        }
        STAMP(1);
        // ---- P = exp2(S*c - LSE*log2e) under the forward's mask and bias ----
        {
            const int col = kw0 + r;
            const bool masked_tile = has_alibi || (kw0 + 32 > sk) || (q0 + 32 > sq) ||
                                     ((p.is_causal || p.is_local) && (kw0 + 32 > q0 + 1 + sk - sq + p.wr)) ||
                                     (p.is_local && (kw0 < q0 + 31 + sk - sq - p.wl));
            if (masked_tile) {
                // Register j of lane (r, hh) is local row roff(j) = (j & 3) + 8 (j >> 2) (a constant)
                // + 4 hh of the sub-block, the lane's key is `col`.  Visible (mask.h:171-172) iff
                //   col <  min(sk, row + 1 + sk - sq + wr)   <=>  roff >= col - shift - wr   (causal / local)
                //   col >= row + sk - sq - wl                 <=>  roff <= col - shift + wl   (local)
                //   row <  sq                                 <=>  roff <  sq - q0 - 4 hh
                // with shift = q0 + 4 hh + sk - sq: one subtract and one unsigned compare per score
                // against the lane's [lo_r, lo_r + span] window.
                constexpr int BIG = 1 << 28;
                const int shift = q0 + 4 * hh + sk - sq;
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
            if constexpr (LSE_IN_ACC) {
#pragma unroll
                for (int j = 0; j < 16; j++) s[j] = __builtin_amdgcn_exp2f(s[j] * c2);
            } else {
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const f32x4 ls = *(const f32x4 *)(lse_s + 8 * g4 + 4 * hh);
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        s[4 * g4 + t] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[4 * g4 + t], c2, -ls[t] * (float)M_LOG2E));
                }
            }
        }
        STAMP(2);
        // ---- dP'[q][key] = dO V^T - delta[q];  dS = P * dP' ----
        f32x16 dp;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const f32x4 dl = *(const f32x4 *)(dl_s + 8 * g4 + 4 * hh);
#pragma unroll
            for (int t = 0; t < 4; t++) dp[4 * g4 + t] = dl[t];   // (-delta)
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const i32x4 da = *(const i32x4 *)(dob + lds_off<DP>(r, 2 * ks + hh));
            i32x4 vb;
            if constexpr (V_IN_LDS) vb = *(const i32x4 *)(vt + lds_off<DP>(32 * wave + r, 2 * ks + hh));
            else vb = vf[ks];
            dp = mma32<BF16>(da, vb, dp);
        }
        STAMP(3);
        // 16-bit B operands: k-step ks2 (16 query rows) = regs 8 ks2 .. 8 ks2 + 7
        i32x4 pf[2], dsf[2];
        unsigned keep = 0xffffu;
        if constexpr (DROP)
            keep = keep_bits_klane(ph_seed, ph_off + (unsigned long long)((bidb * p.h + bidh) * 32), q0 >> 4,
                                   kw0 >> 5, lane, p.drop_keep_u8, xch);
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int j = 8 * ks2 + 2 * w;
                if constexpr (DROP) {
                    // dropped entries: no contribution to dV, and dP' = -delta (dP masked to 0)
                    const bool k0 = (keep >> j) & 1u, k1 = (keep >> (j + 1)) & 1u;
                    const float nd0 = dl_s[8 * (j >> 2) + 4 * hh + (j & 3)];
                    const float nd1 = dl_s[8 * (j >> 2) + 4 * hh + (j & 3) + 1];
                    pf[ks2][w] = pack2<BF16>(k0 ? s[j] : 0.f, k1 ? s[j + 1] : 0.f);
                    dsf[ks2][w] = pack2<BF16>(s[j] * (k0 ? dp[j] : nd0), s[j + 1] * (k1 ? dp[j + 1] : nd1));
                } else {
                    pf[ks2][w] = pack2<BF16>(s[j], s[j + 1]);
                    dsf[ks2][w] = pack2<BF16>(s[j] * dp[j], s[j + 1] * dp[j + 1]);
                }
            }

        // ---- dV^T += dO^T P,  dK^T += Q^T dS  (A operands by transposed reads of the slice) ----
#pragma unroll
        for (int db = 0; db < NDB; db++) {
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ks2++) {
                const int row0 = 16 * ks2 + 4 * hh + tr_q;
                const int ch = 4 * (db0 + db) + 2 * tr_g + (tr_p >> 1);
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
        STAMP(4);
        }  // !skip
        __builtin_amdgcn_sched_barrier(0);  // keep the sub-blocks apart: interleaving them only adds live registers
        if constexpr (SPLIT) {
            if (sub == QSUB - 1 && has_next) {  // between the two sub-blocks
                store_chunk(buf ^ 1, 0, 0);
                load_chunk(nxt_h, nxt_m, 1, 0);
            }
        }
        }  // sub

        if (has_next) {
            if constexpr (DIRECT) store_rowc(buf ^ 1);
            else store_slice(buf ^ 1);
            if constexpr (SPLIT) store_chunk(buf ^ 1, 1, 0);
        }
        if constexpr (DIRECT) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the next slice is in LDS
        STAMP(5);
        __syncthreads();
        STAMP(6);
        buf ^= 1;
        cur_h = nxt_h;
        cur_m = nxt_m;
        advance(nxt_h, nxt_m);
    }
    BLK_STAMP(2);
#ifdef VFA_STAMP
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long *dst = g_stamps_dkdv + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 7; i++) dst[i] = acc_[i];
        dst[7] = (unsigned long long)total_steps;
    }
#endif
    }  // !PP

    if constexpr (HS) {
        // fp32 partials straight from the accumulators: lane (r, hh) holds, per d block and group
        // g4, four consecutive columns of key row kw0 + r.  Slab layout: (rows, h_k, d) contiguous,
        // rows = (batch, key) dense or the packed key rows of a ragged batch.
        if (kw0 + r < sk) {
            const int64_t row = (bi.sum_s_k >= 0 ? (int64_t)bi.sum_s_k : (int64_t)bidb * p.seqlen_k) + kw0 + r;
            const int64_t off = (int64_t)blockIdx.y * p.part_stride + (row * p.h_k + hk) * p.d;
            const float rp = DROP ? p.rp_dropout : 1.f;
#pragma unroll
            for (int db = 0; db < NDB; db++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int col = 32 * (db0 + db) + 8 * g4 + 4 * hh;
                    if (col < p.d) {
                        *(f32x4 *)(p.dk_part + off + col) = f32x4{dk[db][4 * g4] * p.scale_rp, dk[db][4 * g4 + 1] * p.scale_rp,
                                                                  dk[db][4 * g4 + 2] * p.scale_rp, dk[db][4 * g4 + 3] * p.scale_rp};
                        *(f32x4 *)(p.dv_part + off + col) = f32x4{dv[db][4 * g4] * rp, dv[db][4 * g4 + 1] * rp,
                                                                  dv[db][4 * g4 + 2] * rp, dv[db][4 * g4 + 3] * rp};
                    }
                }
        }
        return;
    } else {
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
                    // scale_rp = scale / keep probability (= scale without dropout)
                    w2[0] = pack2<BF16>(dk[db][4 * g4] * p.scale_rp, dk[db][4 * g4 + 1] * p.scale_rp);
                    w2[1] = pack2<BF16>(dk[db][4 * g4 + 2] * p.scale_rp, dk[db][4 * g4 + 3] * p.scale_rp);
                } else {
                    const float rp = DROP ? p.rp_dropout : 1.f;
                    w2[0] = pack2<BF16>(dv[db][4 * g4] * rp, dv[db][4 * g4 + 1] * rp);
                    w2[1] = pack2<BF16>(dv[db][4 * g4 + 2] * rp, dv[db][4 * g4 + 3] * rp);
                }
                *(i32x2 *)(ob + r * Cfg::OUT_ROWB + (32 * (db0 + db) + 8 * g4 + 4 * hh) * 2) = w2;
            }
        // (the staging rows are the wave's own and a wave's LDS operations execute in order: no
        //  barrier between these writes, the reads below and the next round's writes)
        unsigned short *dst = which == 0 ? dkg : dvg;
        const int64_t rs = which == 0 ? p.dk_rs : p.dv_rs;
        constexpr int CPRW = CPR / Cfg::DSPLIT;   // chunks of a row that this workgroup holds
#pragma unroll
        for (int i = 0; i < (32 * CPRW) / 64; i++) {
            const int c = lane + 64 * i;
            const int row = c / CPRW, ch = c % CPRW + 4 * db0;
            const i32x4 w = *(const i32x4 *)(ob + row * Cfg::OUT_ROWB + ch * 16);
            if (kw0 + row < sk && ch * 8 < p.d) *(i32x4 *)(dst + (int64_t)(kw0 + row) * rs + ch * 8) = w;
        }
    }
    }  // !HS
    BLK_STAMP(3);
}

// Sum of the head-split partials -> dk, dv (16 bit), slabs added in index order (deterministic).
// One thread per 8 consecutive d of one (key row, kv head).
template <bool BF16>
__global__ void __launch_bounds__(256)
bwd_kv_reduce_kernel(const KParams p, const int64_t n_rows) {
    const int cpr = p.d / 8;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = n_rows * p.h_k * cpr;
    if (idx >= total) return;
    const int ch = (int)(idx % cpr);
    const int hk = (int)((idx / cpr) % p.h_k);
    const int64_t row = idx / ((int64_t)cpr * p.h_k);
    // row -> (batch, key): dense rows are batch-major; ragged rows are found in cu_seqlens_k
    int bidb, key;
    if (p.cu_seqlens_k) {
        int lo = 0, hi = p.b;  // last b with cu[b] <= row
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (p.cu_seqlens_k[mid] <= row) lo = mid; else hi = mid;
        }
        bidb = lo;
        key = (int)(row - p.cu_seqlens_k[lo]);
    } else {
        bidb = (int)(row / p.seqlen_k);
        key = (int)(row % p.seqlen_k);
    }
    const BlockInfo bi(p, bidb);
    const int64_t off = (row * p.h_k + hk) * p.d + ch * 8;
#pragma unroll
    for (int which = 0; which < 2; which++) {
        const float *src = (which == 0 ? p.dk_part : p.dv_part) + off;
        f32x4 a0 = *(const f32x4 *)src, a1 = *(const f32x4 *)(src + 4);
        for (int s = 1; s < p.hsplit; s++) {
            a0 += *(const f32x4 *)(src + (int64_t)s * p.part_stride);
            a1 += *(const f32x4 *)(src + (int64_t)s * p.part_stride + 4);
        }
        i32x4 w;
        w[0] = pack2<BF16>(a0[0], a0[1]); w[1] = pack2<BF16>(a0[2], a0[3]);
        w[2] = pack2<BF16>(a1[0], a1[1]); w[3] = pack2<BF16>(a1[2], a1[3]);
        unsigned short *dst = which == 0
            ? (unsigned short *)p.dk + bi.k_off(p.dk_bs, p.dk_rs, bidb) + (int64_t)key * p.dk_rs + (int64_t)hk * p.dk_hs
            : (unsigned short *)p.dv + bi.k_off(p.dv_bs, p.dv_rs, bidb) + (int64_t)key * p.dv_rs + (int64_t)hk * p.dv_hs;
        *(i32x4 *)(dst + ch * 8) = w;
    }
}

// Per-(batch) extents that fit a 32-bit byte offset (with room for the tile offsets) take the
// buffer-load paths of the kernels below.
static bool bwd_fits_buffer(const KParams &p) {
    const int64_t lim = (int64_t)1 << 30;
    const int64_t rq = (int64_t)p.seqlen_q + 256, rk = (int64_t)p.seqlen_k + 256;
    return rq * p.q_rs * 2 < lim && rq * p.do_rs * 2 < lim && rk * p.k_rs * 2 < lim && rk * p.v_rs * 2 < lim;
}

template <int DP, bool BF16, int NW, bool FAST, bool DROP, bool HS, bool PP = false, int DV = DP>
static hipError_t launch_dkdv_hs(const KParams &p, hipStream_t stream) {
    using Cfg = BwdCfg<DP, NW>;
    constexpr int LDS = PP ? Cfg::PP_LDS_BYTES : Cfg::LDS_BYTES + (DROP ? NW * 1024 : 0);  // + the Philox exchange space
    const int n_kblocks = (p.seqlen_k + Cfg::KB - 1) / Cfg::KB;
    const long long blocks = (long long)n_kblocks * p.b * p.h_k;
    if (blocks <= 0) return hipSuccess;
    auto kern = bwd_dkdv_kernel<DP, BF16, NW, FAST, DROP, HS, PP, DV>;
    static std::atomic<uint64_t> lds_set{0};
    if (hipError_t e = set_dyn_lds_once((const void *)kern, LDS, lds_set); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)(HS ? p.hsplit : 1), (unsigned)Cfg::DSPLIT), dim3(Cfg::NT), LDS,
                       stream, p, n_kblocks);
    return hipGetLastError();
}

template <int DP, bool BF16, int NW, bool FAST, bool DROP = false, int DV = DP>
static hipError_t launch_dkdv_impl(const KParams &p, hipStream_t stream) {
#ifdef VFA_DEV   // development build only: measured level with the default schedule
    if constexpr (FAST && DP == 128 && NW == 8 && !DROP && DV == DP) {
        // the two-phase schedule (PP in bwd_dkdv_kernel); development switch 0: 3 = always, 4 = never
        // Opt-in: measured level with the default schedule (2.12 vs 2.12 ms on b8 s4096 h32 d128
        // causal, 1.79 vs 1.80 ms on all-zero inputs): beside a partner's vector work the MFMAs
        // issue every ~43 cycles instead of 32, which costs what the overlap saves.
        const bool pp_wanted = g_tuning[0] == 3 || (g_tuning[0] != 4 && VFA_DKDV_PP_DEFAULT);
        // (its loads address a head as the batch entry's base plus a 32-bit byte offset)
        const int64_t lim = (int64_t)1 << 32;
        const bool pp = pp_wanted && (int64_t)p.h * p.q_hs * 2 < lim && (int64_t)p.h * p.do_hs * 2 < lim &&
                        (int64_t)p.h * p.seqlen_q_rounded * 4 < lim;
        if (pp)
            return p.hsplit > 1 ? launch_dkdv_hs<DP, BF16, NW, FAST, DROP, true, true>(p, stream)
                                : launch_dkdv_hs<DP, BF16, NW, FAST, DROP, false, true>(p, stream);
    }
#endif
    return p.hsplit > 1 ? launch_dkdv_hs<DP, BF16, NW, FAST, DROP, true, false, DV>(p, stream)
                        : launch_dkdv_hs<DP, BF16, NW, FAST, DROP, false, false, DV>(p, stream);
}

template <int DP, bool BF16, int NW, bool DROP, int DV = DP>
static hipError_t launch_dkdv(const KParams &p, hipStream_t stream) {
    if constexpr (DROP) {
        return bwd_fits_buffer(p) ? launch_dkdv_impl<DP, BF16, NW, true, true>(p, stream)
                                  : launch_dkdv_impl<DP, BF16, NW, false, true>(p, stream);
    } else {
        return bwd_fits_buffer(p) ? launch_dkdv_impl<DP, BF16, NW, true, false, DV>(p, stream)
                                  : launch_dkdv_impl<DP, BF16, NW, false, false, DV>(p, stream);
    }
}

// ---------------------------------------------------------------------------------------------
// dQ: query-stationary.  Same staging, swizzles and fragment maps as the forward prefill kernel.
constexpr int DQ_BN = 64;

template <int DP, int NW>
struct DqCfg {
    static constexpr int NT = NW * 64;
    static constexpr int BM = NW * 32;
    static constexpr int CPR = DP / 8;
    static constexpr int CH = DQ_BN * CPR / NT;
    static constexpr int TILE_BYTES = DQ_BN * DP * 2;
    static constexpr int O_ROWB = DP * 2 + 16;
    static constexpr int KV_LDS = 4 * TILE_BYTES;
    static constexpr int O_LDS = NW * 32 * O_ROWB;
    // 8 waves, d <= 128: the Q / dO rows of a wave pass through a private LDS image on their way to
    // its fragments (coalesced 1-KiB requests; the fragment layout itself reads 32 bytes per row
    // and request, which keeps the address unit busy for about 4 us per workgroup).  Waves 0 and 1
    // use the second K / V tile buffers, idle until the loop, waves 2.. the space behind the tiles.
    static constexpr bool STAGE_F = NW == 8 && DP <= 128;
    static constexpr int FRAG_BYTES = 2 * 32 * DP * 2;       // Q rows, then dO rows, of one wave
    static constexpr int F_LDS = STAGE_F ? KV_LDS + (NW - 2) * FRAG_BYTES : 0;
    static constexpr int LDS0 = KV_LDS > O_LDS ? KV_LDS : O_LDS;
    static constexpr int LDS_BYTES = LDS0 > F_LDS ? LDS0 : F_LDS;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per workgroup");
};

template <int DP, bool BF16, int NW, bool FAST, bool DROP, int DV = DP>
__global__ void __launch_bounds__(NW * 64, (NW == 8 || DP == 256) ? 1 : 2)
bwd_dq_kernel(const KParams p, const int n_mblocks) {
    using Cfg = DqCfg<DP, NW>;
    constexpr int NT = Cfg::NT, BM = Cfg::BM, CPR = Cfg::CPR, CH = Cfg::CH;
    static_assert(DV % 32 == 0 && DV <= DP && 2 * DV > DP, "DV");   // (see bwd_dkdv_kernel)
    constexpr int KS = DV / 16, NDB = DV / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    DQ_STAMP(0);
    // one word of every 64-byte line of the kernel arguments, requested up front (see the dK/dV kernel)
    int ka_touch = 0;
    {
        const int __attribute__((address_space(4))) *ka =
            (const int __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
#pragma unroll
        for (int i = 0; i < (int)((sizeof(KParams) + 63) / 64); i++) ka_touch |= ka[16 * i];
    }

    // block -> (batch, head, m_block): whole (batch, kv head) pairs per XCD; within an XCD, groups
    // of G pairs with the m-blocks heaviest first across the group's pairs (as in the forward)
    int bidb, bidh, m_block;
    {
        const int g = p.h_ratio;
        const int W = g * n_mblocks;
        const int P = p.b * p.h_k;
        const int L = blockIdx.x;
        const int full = (P / 8) * 8 * W;
        constexpr int SLOTS = 32 * ((NW == 8 || DP == 256) ? 1 : 2);
        int pair, head, mi;
        if (L < full) {
            const int x = L & 7, s = L >> 3;
            const int ppx = P / 8;
            const int G = max(1, min(ppx, (2 * SLOTS + W - 1) / W));
            const int grp = s / (G * W);
            const int within = s - grp * G * W;
            const int Gl = min(G, ppx - grp * G);
            mi = within / (Gl * g);
            const int rest = within - mi * (Gl * g);
            pair = (grp * G + rest / g) * 8 + x;
            head = rest % g;
        } else {
            // the last P % 8 pairs, spread over all XCDs: heaviest m-blocks first across them
            const int Lr = L - full, R = P - (P / 8) * 8;
            mi = Lr / (R * g);
            const int rest = Lr - mi * (R * g);
            pair = (P / 8) * 8 + rest / g;
            head = rest % g;
        }
        bidb = pair / p.h_k;
        bidh = (pair % p.h_k) * g + head;
        m_block = n_mblocks - 1 - mi;
    }
    const int hk = bidh / p.h_ratio;
    const BlockInfo bi(p, bidb);
    const int sq = bi.sq, sk = bi.sk;
    if (m_block * BM >= sq) return;
    DQ_STAMP(4);

    int n_min = !p.is_local ? 0 : max(0, (m_block * BM + sk - sq - p.wl) / DQ_BN);
    int n_max = (sk + DQ_BN - 1) / DQ_BN;
    if (p.is_causal || p.is_local)
        n_max = min(n_max, ((m_block + 1) * BM + sk - sq + p.wr + DQ_BN - 1) / DQ_BN);

    const int q0w = m_block * BM + wave * 32;
    const unsigned short *qg = (const unsigned short *)p.q + bi.q_off(p.q_bs, p.q_rs, bidb) + (int64_t)bidh * p.q_hs;
    const unsigned short *dog = (const unsigned short *)p.dout + bi.q_off(p.do_bs, p.do_rs, bidb) + (int64_t)bidh * p.do_hs;
    unsigned short *dqg = (unsigned short *)p.dq + bi.q_off(p.dq_bs, p.dq_rs, bidb) + (int64_t)bidh * p.dq_hs;

    f32x16 dq[NDB];
#pragma unroll
    for (int i = 0; i < NDB; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) dq[i][j] = 0.f;

    if (n_max > n_min) {
        const unsigned short *kg = (const unsigned short *)p.k + bi.k_off(p.k_bs, p.k_rs, bidb) + (int64_t)hk * p.k_hs;
        const unsigned short *vg = (const unsigned short *)p.v + bi.k_off(p.v_bs, p.v_rs, bidb) + (int64_t)hk * p.v_hs;
        i32x4 kreg[CH], vreg[CH];
        // FAST: as in the forward kernel, K / V tiles through buffer loads bounded at the last key row
        const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)kg, 0, FAST ? (unsigned)(((int64_t)(sk - 1) * p.k_rs + p.d) * 2) : 0u, 0x00020000);
        const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)vg, 0, FAST ? (unsigned)(((int64_t)(sk - 1) * p.v_rs + p.d) * 2) : 0u, 0x00020000);
        const bool colok0 = (tid % CPR) * 8 < p.d;
        const unsigned k_voff = colok0 ? (unsigned)(((tid / CPR) * (int)p.k_rs + (tid % CPR) * 8) * 2) : 0x80000000u;
        const unsigned v_voff = colok0 ? (unsigned)(((tid / CPR) * (int)p.v_rs + (tid % CPR) * 8) * 2) : 0x80000000u;
        const unsigned k_rs2 = (unsigned)p.k_rs * 2, v_rs2 = (unsigned)p.v_rs * 2;
        auto load_tile = [&](int n) {
            if constexpr (FAST) {
                constexpr int RSTEP = NT / CPR;
#pragma unroll
                for (int i = 0; i < CH; i++) {
                    kreg[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        k_rsrc, k_voff + (unsigned)(n * DQ_BN + RSTEP * i) * k_rs2, 0, 0));
                    vreg[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        v_rsrc, v_voff + (unsigned)(n * DQ_BN + RSTEP * i) * v_rs2, 0, 0));
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const int c = tid + i * NT;
                const int row = c / CPR, ch = c % CPR;
                const int grow = n * DQ_BN + row;
                const bool ok = grow < sk && ch * 8 < p.d;
                kreg[i] = ok ? *(const i32x4 *)(kg + (int64_t)grow * p.k_rs + ch * 8) : i32x4{0, 0, 0, 0};
                vreg[i] = ok ? *(const i32x4 *)(vg + (int64_t)grow * p.v_rs + ch * 8) : i32x4{0, 0, 0, 0};
            }
        };
        auto store_tile = [&](int buf) {
            char *kb = smem + buf * Cfg::TILE_BYTES;
            char *vb = smem + (2 + buf) * Cfg::TILE_BYTES;
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const int c = tid + i * NT;
                const int off = lds_off<DP>(c / CPR, c % CPR);
                *(i32x4 *)(kb + off) = kreg[i];
                *(i32x4 *)(vb + off) = vreg[i];
            }
        };
        // DIRECT: the buffer loads write LDS themselves (buffer_load_dwordx4 ... lds), 1 KiB per
        // wave instruction in lane order: each lane fetches the chunk the XOR swizzle places at its
        // slot.  No staging registers, no ds_write; vmcnt(0) before the tile's barrier.
        constexpr bool DIRECT = FAST && DP <= 128;
        constexpr int RPI = 64 / CPR;
        constexpr int NI = DIRECT ? DQ_BN / (RPI * NW) : 1;
        unsigned dk_voff = 0, dv_voff = 0;
        if constexpr (DIRECT) {
            const int row0 = RPI * wave + lane / CPR;
            const int ch = (lane % CPR) ^ swz<DP>(row0);
            const bool ok = ch * 8 < p.d;
            dk_voff = ok ? (unsigned)((row0 * (int)p.k_rs + ch * 8) * 2) : 0x80000000u;
            dv_voff = ok ? (unsigned)((row0 * (int)p.v_rs + ch * 8) * 2) : 0x80000000u;
        }
        auto load_tile_direct = [&](int n, int buf) __attribute__((always_inline)) {
            if constexpr (DIRECT) {
                typedef __attribute__((address_space(3))) void *lds_ptr;
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const unsigned rows = (unsigned)(n * DQ_BN + RPI * NW * i);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        k_rsrc, (lds_ptr)(smem + buf * Cfg::TILE_BYTES + (wave + NW * i) * 1024), 16,
                        dk_voff + rows * k_rs2, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        v_rsrc, (lds_ptr)(smem + (2 + buf) * Cfg::TILE_BYTES + (wave + NW * i) * 1024), 16,
                        dv_voff + rows * v_rs2, 0, 0, 0);
                }
            }
        };

        // The first tile is requested before anything else of the prologue, so that the fragment
        // loads below share its memory round trip.
        if constexpr (DIRECT) load_tile_direct(n_max - 1, 0);
        else load_tile(n_max - 1);

        // Q^T and dO^T B-operand fragments; LSE and delta of this lane's query row
        // (FAST: buffer loads bounded at the last query row -- no predication, the requests go out
        //  back to back)
        i32x4 qf[KS], dof[KS];
        float lse_r = INFINITY, dl_r = 0.f;
        {
            const int row = q0w + r;
            const bool rok = row < sq;
            if constexpr (DIRECT && Cfg::STAGE_F) {
                typedef __attribute__((address_space(3))) void *lds_ptr;
                const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)qg, 0, (unsigned)(((int64_t)(sq - 1) * p.q_rs + p.d) * 2), 0x00020000);
                const auto do_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)dog, 0, (unsigned)(((int64_t)(sq - 1) * p.do_rs + p.d) * 2), 0x00020000);
                char *fb = smem + (wave == 0 ? Cfg::TILE_BYTES : wave == 1 ? 3 * Cfg::TILE_BYTES
                                                               : Cfg::KV_LDS + (wave - 2) * Cfg::FRAG_BYTES);
                // lane -> (row, chunk) of its slot in a 1-KiB request; the swizzle is linear in the
                // row's bits, so the chunk of request i is that of request 0 XOR a constant.  Chunks
                // in the head-dim padding are redirected to the last real one (they meet zero
                // columns of K and V, and finite data stays finite).
                const unsigned col0 = (unsigned)(((lane % CPR) ^ swz<DP>(lane / CPR)) << 4);
                const unsigned col_max = (unsigned)(p.d / 8 - 1) * 16u;
                const unsigned lrow_q = (unsigned)(lane / CPR) * (unsigned)p.q_rs * 2, lrow_do = (unsigned)(lane / CPR) * (unsigned)p.do_rs * 2;
#pragma unroll
                for (int i = 0; i < 32 / RPI; i++) {
                    const unsigned col = min(col0 ^ (unsigned)(swz<DP>(RPI * i) << 4), col_max);
                    const unsigned rws = (unsigned)(q0w + RPI * i);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rsrc, (lds_ptr)(fb + i * 1024), 16,
                                                             col + lrow_q + rws * (unsigned)p.q_rs * 2, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(do_rsrc, (lds_ptr)(fb + 32 * DP * 2 + i * 1024), 16,
                                                             col + lrow_do + rws * (unsigned)p.do_rs * 2, 0, 0, 0);
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the wave's own rows (and the first tile) are in LDS
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    qf[s] = *(const i32x4 *)(fb + lds_off<DP>(r, 2 * s + hh));
                    dof[s] = *(const i32x4 *)(fb + 32 * DP * 2 + lds_off<DP>(r, 2 * s + hh));
                }
                // (the images of waves 0 and 1 are the second tile buffers: read out before the
                //  barrier that precedes the first request into them)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (FAST) {
                const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)qg, 0, (unsigned)(((int64_t)(sq - 1) * p.q_rs + p.d) * 2), 0x00020000);
                const auto do_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)dog, 0, (unsigned)(((int64_t)(sq - 1) * p.do_rs + p.d) * 2), 0x00020000);
                const unsigned qrow = (unsigned)row * (unsigned)p.q_rs * 2, drow = (unsigned)row * (unsigned)p.do_rs * 2;
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const int col = 16 * s + 8 * hh;
                    const bool ok = col < p.d;
                    qf[s] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        q_rsrc, ok ? qrow + (unsigned)col * 2 : 0x80000000u, 0, 0));
                    dof[s] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        do_rsrc, ok ? drow + (unsigned)col * 2 : 0x80000000u, 0, 0));
                }
            } else {
                const unsigned short *qr = qg + (int64_t)(rok ? row : 0) * p.q_rs;
                const unsigned short *dr = dog + (int64_t)(rok ? row : 0) * p.do_rs;
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const int col = 16 * s + 8 * hh;
                    const bool ok = rok && col < p.d;
                    qf[s] = ok ? *(const i32x4 *)(qr + col) : i32x4{0, 0, 0, 0};
                    dof[s] = ok ? *(const i32x4 *)(dr + col) : i32x4{0, 0, 0, 0};
                }
            }
            if (rok) {
                lse_r = p.lse[((int64_t)bidb * p.h + bidh) * p.seqlen_q + row];
                dl_r = p.dsoftmax_sum[((int64_t)bidb * p.h + bidh) * p.seqlen_q_rounded + row];
            }
        }
        const float lse2 = lse_r * (float)M_LOG2E;
        constexpr bool NDL_BLOCK = DP <= 128;   // (at d = 256 the registers are not there)
        f32x16 ndl;
        if constexpr (NDL_BLOCK) {
#pragma unroll
            for (int j = 0; j < 16; j++) ndl[j] = -dl_r;
            asm volatile("" : "+v"(ndl));  // (kept as 16 registers: not rebuilt from dl_r at every use)
        }
        unsigned long long ph_seed = 0, ph_off = 0;
        if constexpr (DROP) {
            ph_seed = p.rng_read ? p.rng_state[0] : p.philox_seed;
            ph_off = (p.rng_read ? p.rng_state[1] : p.philox_offset) +
                     (unsigned long long)((bidb * p.h + bidh) * 32);  // dropout.h:21
        }
        float alibi = 0.f;
        const bool has_alibi = p.alibi_slopes != nullptr;
        if (has_alibi) alibi = p.alibi_slopes[(int64_t)bidb * p.alibi_bs + bidh] / p.scale;
        const float c2 = p.scale_log2;
        const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;

        asm volatile("" ::"s"(ka_touch));
        DQ_STAMP(5);
        if constexpr (DIRECT) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        else store_tile(0);
        DQ_STAMP(6);
        __syncthreads();
        DQ_STAMP(1);
        int buf = 0;
        for (int n = n_max - 1; n >= n_min; n--) {
            const bool has_next = n > n_min;
            if (has_next) {
                if constexpr (DIRECT) load_tile_direct(n - 1, buf ^ 1);
                else load_tile(n - 1);
            }
            const char *kb = smem + buf * Cfg::TILE_BYTES;
            const char *vb = smem + (2 + buf) * Cfg::TILE_BYTES;
            const int c0 = n * DQ_BN;
            bool need = has_alibi || (c0 + DQ_BN > sk);
            if (p.is_causal || p.is_local) need = need || (c0 + DQ_BN > q0w + 1 + sk - sq + p.wr);
            if (p.is_local) need = need || (c0 < q0w + 31 + sk - sq - p.wl);
#pragma unroll
            for (int kb2 = 0; kb2 < 2; kb2++) {
                // skip 32-key halves none of this wave's rows can see (above its diagonal / left
                // of its window); the wave still takes part in the staging
                const int ch0 = c0 + 32 * kb2;
                bool skip = false;
                if (p.is_causal || p.is_local) skip = ch0 >= q0w + 32 + sk - sq + p.wr;
                if (p.is_local) skip = skip || (ch0 + 32 <= q0w + sk - sq - p.wl);
                if (skip) continue;
                // S^T and dP'^T = V dO^T - delta for 32 keys; the query is on the lane
                f32x16 s, dp;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    s[j] = 0.f;
                    if constexpr (!NDL_BLOCK) dp[j] = -dl_r;
                }
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const int off = lds_off<DP>(32 * kb2 + r, 2 * ks + hh);
                    s = mma32<BF16>(*(const i32x4 *)(kb + off), qf[ks], s);
                    // the dP chain starts from the block of -delta (kept for the whole kernel:
                    // filling the accumulator anew would be 16 moves per 32 keys)
                    dp = mma32<BF16>(*(const i32x4 *)(vb + off), dof[ks], (NDL_BLOCK && ks == 0) ? ndl : dp);
                }
                if (need) {
                    const int row = q0w + r;
                    const int lo = p.is_local ? max(0, row + sk - sq - p.wl) : 0;
                    const int hi = (p.is_causal || p.is_local) ? min(sk, row + 1 + sk - sq + p.wr) : sk;
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const int col = c0 + 32 * kb2 + (j & 3) + 8 * (j >> 2) + 4 * hh;
                        float x = s[j];
                        if (has_alibi)
                            x = p.is_causal ? x + alibi * (float)col
                                            : x - alibi * fabsf((float)(row + sk - sq - col));
                        s[j] = (col >= lo && col < hi) ? x : -INFINITY;
                    }
                }
                i32x4 dsf[2];
                unsigned keep = 0xffffu;
                if constexpr (DROP)
                    keep = keep_bits_qlane(ph_seed, ph_off, q0w >> 4, (c0 >> 5) + kb2, r, hh, p.drop_keep_u8);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j], c2, -lse2));
                    if constexpr (DROP) dp[j] = pe * (((keep >> j) & 1u) ? dp[j] : -dl_r);  // dP masked to 0
                    else dp[j] = pe * dp[j];
                }
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
                    for (int w = 0; w < 4; w++)
                        dsf[ks2][w] = pack2<BF16>(dp[8 * ks2 + 2 * w], dp[8 * ks2 + 2 * w + 1]);
                // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
                for (int db = 0; db < NDB; db++) {
#pragma unroll
                    for (int ks2 = 0; ks2 < 2; ks2++) {
                        const int row0 = 32 * kb2 + 16 * ks2 + 4 * hh + tr_q;
                        const int ch = 4 * db + 2 * tr_g + (tr_p >> 1);
                        const i32x2 a0 = __builtin_bit_cast(i32x2, lds_read_tr16(kb + lds_off<DP>(row0, ch) + 8 * (tr_p & 1)));
                        const i32x2 a1 = __builtin_bit_cast(i32x2, lds_read_tr16(kb + lds_off<DP>(row0 + 8, ch) + 8 * (tr_p & 1)));
                        const i32x4 kt = {a0[0], a0[1], a1[0], a1[1]};
                        dq[db] = mma32<BF16>(kt, dsf[ks2], dq[db]);
                    }
                }
            }
            if constexpr (DIRECT) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) before the barrier
            else if (has_next) store_tile(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---- dq = round16(scale * dQ), stored through LDS as whole rows ----
    DQ_STAMP(2);
    __syncthreads();
    char *ob = smem + wave * 32 * Cfg::O_ROWB;
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
        if (q0w + row < sq && ch * 8 < p.d)
            *(i32x4 *)(dqg + (int64_t)(q0w + row) * p.dq_rs + ch * 8) = w;
    }
    DQ_STAMP(3);
}

template <int DP, bool BF16, int NW, bool FAST, bool DROP = false, int DV = DP>
static hipError_t launch_dq_impl(const KParams &p, hipStream_t stream) {
    using Cfg = DqCfg<DP, NW>;
    const int n_mblocks = (p.seqlen_q + Cfg::BM - 1) / Cfg::BM;
    const long long blocks = (long long)n_mblocks * p.b * p.h;
    if (blocks <= 0) return hipSuccess;
    auto kern = bwd_dq_kernel<DP, BF16, NW, FAST, DROP, DV>;
    static std::atomic<uint64_t> lds_set{0};
    if (hipError_t e = set_dyn_lds_once((const void *)kern, Cfg::LDS_BYTES, lds_set); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p, n_mblocks);
    return hipGetLastError();
}

template <int DP, bool BF16, int NW, int DV = DP>
static hipError_t launch_dq(const KParams &p, hipStream_t stream) {
    return bwd_fits_buffer(p) ? launch_dq_impl<DP, BF16, NW, true, false, DV>(p, stream)
                              : launch_dq_impl<DP, BF16, NW, false, false, DV>(p, stream);
}

hipError_t run_bwd_dkdv_w64(const KParams &p, hipStream_t stream);  // bwd_dkdv_w64.hip
#if !defined(VFA_DROPOUT_TU) && !defined(VFA_DQ_ILP_TU) && !defined(VFA_BWD96_TU)
// Whether the 64-keys-per-wave dK/dV kernel can run this problem.
bool bwd_dkdv_w64_ok(const KParams &p) {
    return p.d <= 128 && p.hsplit <= 1 && p.seqlen_k > 128 && bwd_fits_buffer(p);
}
bool bwd_dkdv_wide_ok(const KParams &p) { return p.d > 128 && p.hsplit <= 1 && bwd_fits_buffer(p); }
#endif
hipError_t run_bwd_dkdv_wide(const KParams &p, hipStream_t stream);  // bwd_dkdv_wide.hip
hipError_t run_bwd_dq_ilp(const KParams &p, hipStream_t stream, bool waves8);   // bwd_dq_ilp.hip (this file, other compiler flags)
bool bwd_dkdv_persist_ok(const KParams &p);                          // bwd_dkdv_persist.hip
long long bwd_dkdv_persist_items(const KParams &p);
hipError_t run_bwd_dkdv_persist(const KParams &p, hipStream_t stream);
bool bwd_dq_persist_ok(const KParams &p);                            // bwd_dq_persist.hip
long long bwd_dq_persist_items(const KParams &p);
hipError_t run_bwd_dq_persist(const KParams &p, hipStream_t stream);

template <int DP, bool BF16, bool DROP, int DV = DP>
static hipError_t launch_bwd(const KParams &p, hipStream_t stream, hipEvent_t *ev) {
    hipError_t e;
    if (ev && (e = hipEventRecord(ev[0], stream)) != hipSuccess) return e;
    {
        const int64_t rows = (int64_t)p.b * p.h * p.seqlen_q;
        hipLaunchKernelGGL(bwd_preprocess_kernel<BF16>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, p);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (ev && (e = hipEventRecord(ev[1], stream)) != hipSuccess) return e;
    // the geometry: vfa_dispatch.h (work per CU first, keys per block second).  The 64-keys-per-wave
    // kernel (bwd_dkdv_w64.hip) and this file's kernels produce the same bits.
    const int geo_kv = DROP ? (DP == 256 ? VFA_GEO_BLOCK4 : VFA_GEO_BLOCK8) : dkdv_geometry(p, device_num_cus());
    bool persist_kv = false;
#ifdef VFA_DEV
    // Persistent dK/dV workgroups (bwd_dkdv_persist.hip): development build only, switch 0 = 6.
    // Correct, but 11-18 % behind this kernel at seqlen 4096 and 20 % at 512: wrapped in the walk,
    // the step loop comes out of the compiler with every LDS read directly in front of its MFMA,
    // under every scheduling strategy of hipcc 7.2 that does not spill.
    if constexpr (!DROP && DP == 128) persist_kv = g_tuning[0] == 6 && bwd_dkdv_persist_ok(p);
    if (persist_kv) e = run_bwd_dkdv_persist(p, stream);
#endif
    if (persist_kv) {}
    else if (geo_kv == VFA_GEO_WAVE64) e = run_bwd_dkdv_w64(p, stream);
    else if (geo_kv == VFA_GEO_WIDE32) e = run_bwd_dkdv_wide(p, stream);
    else if constexpr (DP == 256) e = launch_dkdv<DP, BF16, 4, DROP>(p, stream);
    else e = geo_kv == VFA_GEO_BLOCK4 ? launch_dkdv<DP, BF16, 4, DROP, DV>(p, stream) : launch_dkdv<DP, BF16, 8, DROP, DV>(p, stream);
    if (e != hipSuccess) return e;
    if (p.hsplit > 1) {
        const int64_t items = p.part_rows * p.h_k * (p.d / 8);
        hipLaunchKernelGGL(bwd_kv_reduce_kernel<BF16>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, p,
                           p.part_rows);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (ev && (e = hipEventRecord(ev[2], stream)) != hipSuccess) return e;
    {
        // the geometry: vfa_dispatch.h.  Persistent workgroups (bwd_dq_persist.hip) wherever every CU
        // gets at least two 256-row blocks to chain; else 8 waves (256 query rows per workgroup) or two
        // 4-wave workgroups per CU, which overlap each other's prologues.
        const int geo_q = DROP ? VFA_GEO_BLOCK4 : dq_geometry(p, device_num_cus());
        if (geo_q == VFA_GEO_PERSIST8) e = run_bwd_dq_persist(p, stream);
        else if constexpr (DROP) e = bwd_fits_buffer(p) ? launch_dq_impl<DP, BF16, 4, true, true>(p, stream)
                                                        : launch_dq_impl<DP, BF16, 4, false, true>(p, stream);
        else if constexpr (DP >= 128) e = run_bwd_dq_ilp(p, stream, geo_q == VFA_GEO_BLOCK8);
        else e = geo_q == VFA_GEO_BLOCK8 ? launch_dq<DP, BF16, 8>(p, stream) : launch_dq<DP, BF16, 4>(p, stream);
        if (e != hipSuccess) return e;
    }
    if (ev && (e = hipEventRecord(ev[3], stream)) != hipSuccess) return e;
    return hipGetLastError();
}

#if defined(VFA_DQ_ILP_TU)
// The one-block-per-workgroup dQ kernels from d = 128 up, a translation unit of their own
// (bwd_dq_ilp.hip) because they are built under LLVM's ILP-first scheduler: the default one puts
// every LDS read right in front of its MFMA, which nothing hides at one wave per SIMD (d > 128:
// measured b4 s4096 h16 d256 1.075 -> 0.953 ms causal, 1.959 -> 1.734 full) and little does at two
// (d = 128 with the persistent kernel switched off: b8 s4096 h32 causal 1.500 -> 1.443 ms, b32 s512
// h16 0.104 -> 0.097).  At d = 64 the two schedulers tie (-3 ... +1.5 %), and the same flag makes the
// dK/dV kernels of this file spill.
hipError_t run_bwd_dq_ilp(const KParams &p, hipStream_t stream, bool waves8) {
    const bool bf = p.is_bf16;
    if (p.d > 128) {
        // head dims up to 160 / 192 stop their products at that column
        if (p.d <= 160) return bf ? launch_dq<256, true, 4, 160>(p, stream) : launch_dq<256, false, 4, 160>(p, stream);
        if (p.d <= 192) return bf ? launch_dq<256, true, 4, 192>(p, stream) : launch_dq<256, false, 4, 192>(p, stream);
        return bf ? launch_dq<256, true, 4>(p, stream) : launch_dq<256, false, 4>(p, stream);
    }
    if (p.d <= 96) {   // products cut off at column 96 (see run_bwd_96)
        if (waves8) return bf ? launch_dq<128, true, 8, 96>(p, stream) : launch_dq<128, false, 8, 96>(p, stream);
        return bf ? launch_dq<128, true, 4, 96>(p, stream) : launch_dq<128, false, 4, 96>(p, stream);
    }
    if (waves8) return bf ? launch_dq<128, true, 8>(p, stream) : launch_dq<128, false, 8>(p, stream);
    return bf ? launch_dq<128, true, 4>(p, stream) : launch_dq<128, false, 4>(p, stream);
}
#elif defined(VFA_BWD96_TU)
// Head dims 65..96, a translation unit of their own (bwd_d96.hip): the d = 128 layouts with the
// products cut off at column 96 -- 6 of 8 k-steps of S and dP, 3 of 4 column blocks of dK, dV and dQ;
// the skipped ones multiply zero padding, so the gradients are the same bits.  Measured b8 s4096 h32
// d96 causal: dK/dV 1.937 -> 1.666 ms, dQ 1.413 -> 1.235.
hipError_t run_bwd_96(const KParams &p, hipStream_t stream, hipEvent_t *ev) {
    return p.is_bf16 ? launch_bwd<128, true, false, 96>(p, stream, ev) : launch_bwd<128, false, false, 96>(p, stream, ev);
}
#elif !defined(VFA_DROPOUT_TU)
hipError_t run_bwd_96(const KParams &p, hipStream_t stream, hipEvent_t *ev);   // bwd_d96.hip
hipError_t run_bwd(const KParams &p, hipStream_t stream, hipEvent_t *ev) {
    const bool bf = p.is_bf16;
    if (p.d > 64 && p.d <= 96) return run_bwd_96(p, stream, ev);
    if (p.d <= 64) return bf ? launch_bwd<64, true, false>(p, stream, ev) : launch_bwd<64, false, false>(p, stream, ev);
    if (p.d <= 128) return bf ? launch_bwd<128, true, false>(p, stream, ev) : launch_bwd<128, false, false>(p, stream, ev);
    return bf ? launch_bwd<256, true, false>(p, stream, ev) : launch_bwd<256, false, false>(p, stream, ev);
}
#else   // the dropout variants are compiled as a translation unit of their own (bwd_drop.hip)
hipError_t run_bwd_dropout(const KParams &p, hipStream_t stream, hipEvent_t *ev) {
    const bool bf = p.is_bf16;
    if (p.d <= 64) return bf ? launch_bwd<64, true, true>(p, stream, ev) : launch_bwd<64, false, true>(p, stream, ev);
    if (p.d <= 128) return bf ? launch_bwd<128, true, true>(p, stream, ev) : launch_bwd<128, false, true>(p, stream, ev);
    return bf ? launch_bwd<256, true, true>(p, stream, ev) : launch_bwd<256, false, true>(p, stream, ev);
}
#endif

}  // namespace vfa

#if defined(VFA_STAMP) && !defined(VFA_DROPOUT_TU) && !defined(VFA_DQ_ILP_TU) && !defined(VFA_BWD96_TU)
extern "C" __attribute__((visibility("default"))) int vfa_debug_read_stamps_dkdv(unsigned long long *host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vfa::g_stamps_dkdv), n * sizeof(unsigned long long));
}
extern "C" __attribute__((visibility("default"))) int vfa_debug_read_blk_dkdv(unsigned long long *host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vfa::g_blk_dkdv), n * sizeof(unsigned long long));
}
extern "C" __attribute__((visibility("default"))) int vfa_debug_read_blk_dq(unsigned long long *host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vfa::g_blk_dq), n * sizeof(unsigned long long));
}
#endif
