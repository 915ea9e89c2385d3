// Forward kernel for head dims above 128 (DP = 256): 4 waves at ONE wave per SIMD, 32 query rows per
// wave, contiguous K/V, no dropout / ALiBi -- the tile loop of fwd_prefill_kernel
// (flash_fwd_kernel.h:27-470) software-pipelined by hand.
//
// At this head dim a wave's O^T accumulator is 128 registers and its Q fragments 64: one wave per
// SIMD with the 512-entry register file.  Nothing but the wave's own instruction stream then hides
// its LDS reads and its vector work, and fwd_prefill_kernel -- written for two or three waves per
// SIMD, every LDS read right in front of its MFMA -- leaves the matrix pipe idle for most of a tile
// (measured b4 s4096 h16 d256: 690 / 750 TFLOP/s causal / full).  Here
//
//   iteration i:  phase 1   S(i) = K(i) Q^T            (32 MFMAs)  | second half of softmax(i-1), the loads
//                 phase 2   O += V(i-1)^T P(i-1)^T     (32 MFMAs)  | mask / row max of S(i), first half of softmax(i)
//
// S is double-buffered in the arch VGPRs; O (128 registers) and the Q fragments (64) live in the
// accumulator half of the register file at FIXED registers (a[0:127], a[128:191]) that only the asm
// statements of fwd_w64_regs.inc touch (the register map of the 64-rows-per-wave kernel at d = 128:
// 8 O^T tiles, 16 Q fragments): values the compiler allocates there are copied between the two files
// at every branch join.  The MFMAs are therefore opaque to the compiler; the wait states between
// them and the vector instructions that feed or read them are placed by hand (operand_fence, the
// s_nop groups).  K(i+2) and V(i) are fetched straight into LDS (asm-issued, so that the compiler's
// wait-count pass does not hold LDS reads behind them) during phase 1, one barrier per iteration.
// LDS: a ring of three K tiles and two V tiles of 32 KiB = 160 KiB, one workgroup per CU.
//
// This translation unit is compiled with -mllvm -amdgpu-mfma-vgpr-form=1 (MFMA results of the S
// products in VGPRs: by default a kernel that may use more than 256 registers gets them in AGPRs).
#include "vfa_kernel_params.h"
#include "vfa_one_wave_helpers.h"
#include <type_traits>

namespace vfa {

#ifndef WIDE_RING
#define WIDE_RING 3
#endif

namespace fw256 {
constexpr int DP = 256, BN = 64, NW = 4, QM = 32, BM = NW * QM;
constexpr int CPR = DP / 8, ROWB = DP * 2, TILE = BN * ROWB;
// (k-steps of Q K^T and 32-column blocks of O: per instantiation, KS = DV / 16 and NDB = DV / 32)
constexpr int O_ROWB = DP * 2 + 16;
constexpr int KV_LDS = 5 * TILE;                 // K0 K1 K2 V0 V1
constexpr int O_LDS = NW * QM * O_ROWB;
constexpr int LDS_BYTES = KV_LDS > O_LDS ? KV_LDS : O_LDS;
}  // namespace fw256

#include "fwd_w64_regs.inc"

// max(a, b, c) as one instruction (fmaxf on values the compiler cannot prove canonical costs an
// extra v_max per operand)
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r_;
    // (volatile: keeps its place among the MFMA statements, which is what separates the first read
    // of an S tile from the MFMAs that wrote it)
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r_) : "v"(a), "v"(b), "v"(c));
    return r_;
}

// Makes the compiler wait HERE for the LDS read that fills `x` (it puts its s_waitcnt in front of the
// first statement that names the register).  Used right behind an MFMA pair for the fragment of the
// NEXT pair: an MFMA statement that directly follows the wait for its ds_read_b64_tr_b16 operand read
// stale data on gfx950 (a wait-state requirement the compiler only knows for MFMAs it emits
// itself; it showed as NaNs that came and went with the code's alignment).  With the wait moved up,
// the vector work of the step sits between the wait and the MFMA.
__device__ __forceinline__ void ready(i32x4 &x) { asm volatile("s_nop 0" : "+v"(x)); }
// The same for the paths that are not software-pipelined: wait, then four wait states.
__device__ __forceinline__ void settle(i32x4 &x) { asm volatile("s_nop 3" : "+v"(x)); }

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// PAGED: K/V through a block table whose pages are a power of two and a multiple of 16 rows.  A wave
// then fills 16 consecutive rows of a tile -- they lie in one page and share a descriptor, built in
// scalar registers from a page id that was requested one iteration earlier -- and a lane's chunk
// in instruction i is its chunk in instruction 0 XOR a constant (the swizzle is linear in the
// row's bits).  Chunks in the head-dim padding are redirected to the last real chunk: they meet zero
// columns of Q (K) or columns of O that are never stored (V), and finite data stays finite.
template <bool BF16, bool PAGED, int DV = fw256::DP>
__global__ void __launch_bounds__(256, 1)
fwd_prefill_wide_kernel(const KParams p, const int n_mblocks) {
    using namespace fw256;
    // DV <= DP: the head dims this instantiation serves, rounded up to 32.  Tiles, Q fragments and
    // the O registers keep the layout of DP columns; the products stop at column DV (what is skipped
    // multiplies zero padding), and the vector work and loads that ran beside the skipped MFMAs are
    // spread over the remaining ones.
    static_assert(DV % 32 == 0 && DV <= DP && DV > 128, "DV");
    constexpr int KS = DV / 16, NDB = DV / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;

    int bidb, bidh, m_block;
    map_block(p, n_mblocks, 32, bidb, bidh, m_block);
    const int hk = bidh / p.h_ratio;
    const BlockInfo bi(p, bidb);
    const int sq = bi.sq, sk = bi.sk;
    if (m_block * BM >= sq) return;

    // key tile range of the row block: flash_fwd_kernel.h:59-67
    const int n_min = !p.is_local ? 0 : max(0, (m_block * BM + sk - sq - p.wl) / BN);
    int n_max = (sk + BN - 1) / BN;
    if (p.is_causal || p.is_local) n_max = min(n_max, ((m_block + 1) * BM + sk - sq + p.wr + BN - 1) / BN);

    const unsigned short *qg = (const unsigned short *)p.q + bi.q_off(p.q_bs, p.q_rs, bidb) + (int64_t)bidh * p.q_hs;
    unsigned short *og = (unsigned short *)p.o + bi.q_off(p.o_bs, p.o_rs, bidb) + (int64_t)bidh * p.o_hs;
    float *lse_g = p.lse + ((int64_t)bidb * p.h + bidh) * p.seqlen_q;
    const int q0w = m_block * BM + wave * QM;  // first query row of this wave

    if (n_max <= n_min) {  // no visible key: O = 0, LSE = +inf (flash_fwd_kernel.h:70-106)
        for (int c = tid; c < BM * CPR; c += 256) {
            const int row = m_block * BM + c / CPR, ch = c % CPR;
            if (row < sq && ch * 8 < p.d) *(i32x4 *)(og + (int64_t)row * p.o_rs + ch * 8) = i32x4{0, 0, 0, 0};
        }
        for (int c = tid; c < BM; c += 256)
            if (m_block * BM + c < sq) lse_g[m_block * BM + c] = INFINITY;
        return;
    }

    // ---- K/V: contiguous per (batch, head), fetched straight into LDS ----
    const int bidb_cache = p.cache_batch_idx ? p.cache_batch_idx[bidb] : bidb;
    const unsigned short *kg = (const unsigned short *)p.k + (int64_t)hk * p.k_hs;
    const unsigned short *vg = (const unsigned short *)p.v + (int64_t)hk * p.v_hs;
    const int *bt = nullptr;
    if constexpr (PAGED) {
        bt = p.block_table + (int64_t)bidb * p.bt_bs;
    } else {
        kg += bi.k_off(p.k_bs, p.k_rs, bidb_cache);
        vg += bi.k_off(p.v_bs, p.v_rs, bidb_cache);
    }
    const i32x4 k_rsrc = make_rsrc(kg, PAGED ? 0u : (unsigned)(((int64_t)(sk - 1) * p.k_rs + p.d) * 2));
    const i32x4 v_rsrc = make_rsrc(vg, PAGED ? 0u : (unsigned)(((int64_t)(sk - 1) * p.v_rs + p.d) * 2));
    const i32x4 k_rsrc_none = make_rsrc(kg, 0u);  // every lane out of range: the load writes zeros
    // A wave instruction fills 1 KiB of LDS in lane order = 2 rows of the tile; lane -> the chunk the
    // XOR swizzle places at its slot.  The rows of a wave's instructions are 8 apart: the lane's
    // swizzle alternates between two values, for the even and the odd instructions.  Out-of-range
    // lanes write zeros.
    constexpr int RPI = 64 / CPR, NI = BN / (RPI * NW);
    static_assert(RPI * NW == 8 && NI == 8, "rounds of 8 rows");
    unsigned dk_voff[2], dv_voff[2];
    {
        const int row0 = RPI * wave + lane / CPR;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int ch = (lane % CPR) ^ swz<DP>(row0 + 8 * e);
            const bool ok = ch * 8 < p.d;
            dk_voff[e] = ok ? (unsigned)((row0 * (int)p.k_rs + ch * 8) * 2) : 0x80000000u;
            dv_voff[e] = ok ? (unsigned)((row0 * (int)p.v_rs + ch * 8) * 2) : 0x80000000u;
        }
    }
    const unsigned k_rs2 = (unsigned)p.k_rs * 2, v_rs2 = (unsigned)p.v_rs * 2;
    const unsigned lds0 = 0;  // the dynamic LDS segment starts at LDS address 0 (no static LDS here)
    // PAGED: the lane's column in instruction 0 of its wave's 16-row group, and its row part
    const unsigned pcol0 = (unsigned)(((lane % CPR) ^ swz<DP>(lane / CPR)) << 4);
    const unsigned pcol_max = (unsigned)(p.d / 8 - 1) * 16u;
    const unsigned prow_k = (unsigned)(lane / CPR) * k_rs2, prow_v = (unsigned)(lane / CPR) * v_rs2;
    // page ids (scalar) of this wave's row group in tiles n, n - 1, n - 2 of the current iteration
    int pq0 = 0, pq1 = 0, pq2 = 0;
    auto page_of = [&](const int n) __attribute__((always_inline)) -> int {
        const int rb = n * BN + 16 * wave;
        return bt[(rb >= 0 && rb < sk) ? rb >> p.page_shift : 0];
    };
    auto page_rsrc = [&](const unsigned short *base, const int64_t bs, const int64_t rs, const int pg, const int n,
                         const bool exists) __attribute__((always_inline)) -> i32x4 {
        const int rb = n * BN + 16 * wave;
        const int valid = exists ? min(sk - rb, 16) : 0;  // rows of the group that exist
        return make_rsrc(base + (int64_t)pg * bs + (int64_t)(rb & (p.page_size - 1)) * rs,
                         valid > 0 ? (unsigned)(((int64_t)(valid - 1) * rs + p.d) * 2) : 0u);
    };
    // The descriptors and row bases of an iteration's loads: K tile n - 2 (none: the empty descriptor,
    // no branch in the loop) and V tile n.
    struct Loads { i32x4 kr, vr; unsigned kbase, vbase; };
    auto begin_loads = [&](const int it) __attribute__((always_inline)) -> Loads {
        const int n = n_max - 1 - it;
        const bool k_next = n - 2 >= n_min;
        Loads L;
        if constexpr (PAGED) {
            L.kr = page_rsrc(kg, p.k_bs, p.k_rs, pq2, n - 2, k_next);
            L.vr = page_rsrc(vg, p.v_bs, p.v_rs, pq0, n, true);
            L.kbase = 0;
            L.vbase = 0;
            pq0 = pq1;
            pq1 = pq2;
            pq2 = page_of(n - 3);  // (used in the next iteration)
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) L.kr[t] = k_next ? k_rsrc[t] : k_rsrc_none[t];
            L.vr = v_rsrc;
            L.kbase = (unsigned)((n - 2) * BN) * k_rs2;
            L.vbase = (unsigned)(n * BN) * v_rs2;
        }
        return L;
    };
    auto pcol = [&](const int i) __attribute__((always_inline)) -> unsigned {
        return min(pcol0 ^ (unsigned)(swz<DP>(RPI * i) << 4), pcol_max);
    };
    auto dma_k = [&](const Loads &L, const int slot, const int i) __attribute__((always_inline)) {
        if constexpr (PAGED)
            dma_16b_to_lds(L.kr, pcol(i) + prow_k + (unsigned)(RPI * i) * k_rs2, lds0 + slot * TILE + wave * 8192 + i * 1024);
        else
            dma_16b_to_lds(L.kr, dk_voff[i & 1] + L.kbase + (unsigned)(RPI * NW * i) * k_rs2, lds0 + slot * TILE + (wave + NW * i) * 1024);
    };
    auto dma_v = [&](const Loads &L, const int slot, const int i) __attribute__((always_inline)) {
        if constexpr (PAGED)
            dma_16b_to_lds(L.vr, pcol(i) + prow_v + (unsigned)(RPI * i) * v_rs2, lds0 + (3 + slot) * TILE + wave * 8192 + i * 1024);
        else
            dma_16b_to_lds(L.vr, dv_voff[i & 1] + L.vbase + (unsigned)(RPI * NW * i) * v_rs2, lds0 + (3 + slot) * TILE + (wave + NW * i) * 1024);
    };
    // a K tile outside the pipeline (the prologue's two)
    auto load_k_now = [&](const int n, const int slot) __attribute__((always_inline)) {
        Loads L;
        if constexpr (PAGED) {
            L.kr = page_rsrc(kg, p.k_bs, p.k_rs, page_of(n), n, true);
            L.kbase = 0;
        } else {
            L.kr = k_rsrc;
            L.kbase = (unsigned)(n * BN) * k_rs2;
        }
#pragma unroll
        for (int i = 0; i < NI; i++) dma_k(L, slot, i);
    };

    static_for<NDB>([&](auto T) { o_zero<decltype(T)::value>(); });  // O^T tile db
    float m_run = -INFINITY;  // reference point of the exponentials, followed lazily
    float l_run = 0.f;        // running row sum, partial per lane half
    const float c2 = p.scale_log2;

    // Per-lane LDS read offsets.  The swizzle touches the low four bits of the chunk index only, so
    // k-step ks + 8 and d block db + 4 are 256 bytes further on.
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
    int k_rd8[8], v_rd4[4][2];
#pragma unroll
    for (int ks = 0; ks < 8; ks++) k_rd8[ks] = lds_off<DP>(r, 2 * ks + hh);
#pragma unroll
    for (int db = 0; db < 4; db++)
#pragma unroll
        for (int e = 0; e < 2; e++)
            v_rd4[db][e] = lds_off<DP>(4 * hh + tr_q + 8 * e, 4 * db + 2 * tr_g + (tr_p >> 1)) + 8 * (tr_p & 1);
    auto k_rd = [&](const int ks) __attribute__((always_inline)) { return k_rd8[ks & 7] + 256 * (ks >> 3); };
    auto v_rd = [&](const int db, const int e) __attribute__((always_inline)) { return v_rd4[db & 3][e] + 256 * (db >> 2); };

    // Tiles this wave's rows see, a sub-range of the block's: the block's first iterations cover
    // tiles above the diagonal of the lower waves' rows; those waves only help with the loads.
    int n_top = n_max;
    if (p.is_causal || p.is_local) n_top = min(n_max, (q0w + QM + sk - sq + p.wr + BN - 1) / BN);
    if (q0w >= sq) n_top = n_min;  // a wave beyond the last row
    n_top = max(n_top, n_min);
    const int n_it = n_max - n_min;           // block iterations (tile n_max - 1 - it)
    const int skip = n_max - n_top;           // leading iterations without work for this wave

    f32x16 sA[2], sB[2];        // S^T tiles [key half], two pipeline stages
    i32x4 pf[4];                // P^T fragments [k-step of 16 keys] of the tile in flight
    float mc = 0.f;             // reference point times scale_log2 (of the tile whose P is being built)
    float nmc = 0.f;            // its negative (the addend of the exponent's multiply-add)
    float alpha_pend = 1.f;
    bool rescale_pend = false;

    // ---- pieces of one tile's work ----
    // masks of tile n (wave-uniform test outside): causal / local / key tail (mask.h:128-209)
    auto mask_tile = [&](f32x16 (&s)[2], const int n) __attribute__((always_inline)) {
        const int c0 = n * BN;
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("" : "+v"(s[kh][j]));  // reads stay behind the MFMAs before this point
        const int row = q0w + r;
        const int lo = p.is_local ? max(0, row + sk - sq - p.wl) : 0;
        const int hi = (p.is_causal || p.is_local) ? min(sk, row + 1 + sk - sq + p.wr) : sk;
        // register j of key half kh is key c0 + 32 kh + (j & 3) + 8 (j >> 2) + 4 hh
        const int lo_r = lo - c0 - 4 * hh, hi_r = hi - c0 - 4 * hh;
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int off = 32 * kh + (j & 3) + 8 * (j >> 2);
                s[kh][j] = (off >= lo_r && off < hi_r) ? s[kh][j] : -INFINITY;
            }
    };
    auto needs_mask = [&](const int n) {
        const int c0 = n * BN;
        bool need = c0 + BN > sk;
        if (p.is_causal || p.is_local) need = need || (c0 + BN > q0w + 1 + sk - sq + p.wr);
        if (p.is_local) need = need || (c0 < q0w + QM - 1 + sk - sq - p.wl);
        return need;
    };
    // the (lazy) move of the reference point given the tile's row maximum: softmax.h:135-166
    constexpr float RESCALE_LOG2 = 8.f;
    auto move_reference = [&](const float mx) __attribute__((always_inline)) {
        const bool upd = __any((mx - m_run) * c2 > RESCALE_LOG2);
        if (upd) {
            const float m_new = fmaxf(m_run, mx);
            const float m_cur = (m_new == -INFINITY) ? 0.f : m_new;
            const float al = __builtin_amdgcn_exp2f((m_run - m_cur) * c2);
            l_run *= al;
            alpha_pend = al;   // O is rescaled once the P V products in flight have landed
            m_run = m_new;
            rescale_pend = true;
        }
        mc = ((m_run == -INFINITY) ? 0.f : m_run) * c2;
        nmc = -mc;
    };
    auto row_max = [&](f32x16 (&s)[2]) __attribute__((always_inline)) {
        float a = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; j++) a = max3(a, s[0][j], s[1][j]);
        move_reference(half_swap_max(a));
    };
    // two exponentials of the tile in `s_`: P dword W of k-step E2 of key half KH.  The scores come
    // out of asm MFMAs: their first read is a volatile statement, which keeps its place among the MFMA
    // statements (the compiler would otherwise move the read up to right behind the MFMA that
    // produced the score, before the result has landed).
    auto exp_pair = [&](f32x16 (&s_)[2], auto KH, auto E2, auto W) __attribute__((always_inline)) {
        constexpr int kh = decltype(KH)::value, e2 = decltype(E2)::value, w = decltype(W)::value;
        constexpr int j = 8 * e2 + 2 * w;
        float x0, x1;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x0) : "v"(s_[kh][j]), "s"(c2), "v"(nmc));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x1) : "v"(s_[kh][j + 1]), "s"(c2), "v"(nmc));
        const float e0 = __builtin_amdgcn_exp2f(x0);
        const float e1 = __builtin_amdgcn_exp2f(x1);
        l_run += e0 + e1;
        pf[2 * kh + e2][w] = pack2<BF16>(e0, e1);
    };
    // all eight pairs of key half KH
    auto exp_half = [&](f32x16 (&s_)[2], auto KH) __attribute__((always_inline)) {
        static_for<8>([&](auto I) {
            constexpr int i = decltype(I)::value;
            exp_pair(s_, KH, std::integral_constant<int, (i >> 2)>{}, std::integral_constant<int, (i & 3)>{});
        });
    };
    // V fragment (transposed read) G = 8 ks + db of the V tile at `vb`
    auto v_frag = [&](const char *vb, auto G) __attribute__((always_inline)) -> i32x4 {
        constexpr int ks = decltype(G)::value / NDB, db = decltype(G)::value % NDB;
        const i32x2 a0 = __builtin_bit_cast(i32x2, lds_read_tr16(vb + v_rd(db, 0) + 16 * ks * ROWB));
        const i32x2 a1 = __builtin_bit_cast(i32x2, lds_read_tr16(vb + v_rd(db, 1) + 16 * ks * ROWB));
        return i32x4{a0[0], a0[1], a1[0], a1[1]};
    };
    // O^T += V^T P^T, whole tile, nothing beside it (mask-edge tiles and the drain)
    auto pv_tile = [&](const char *vb) __attribute__((always_inline)) {
        static_for<4 * NDB>([&](auto G) {
            constexpr int g = decltype(G)::value;
            i32x4 vf = v_frag(vb, G);
            settle(vf);
            mma_o<g % NDB, BF16>(vf, pf[g / NDB]);
        });
    };
    auto apply_rescale = [&]() __attribute__((always_inline)) {
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // the last P V results have landed
        static_for<NDB>([&](auto T) { o_scale<decltype(T)::value>(alpha_pend); });
        asm volatile("s_nop 3" ::: "memory");
        rescale_pend = false;
    };
    // K tiles go through a ring of three LDS slots (tile of iteration it in slot it % 3, fetched two
    // iterations ahead), V tiles through two (fetched in the iteration of their S, used by the P V of
    // the next one).
    auto issue_loads = [&](const int it) __attribute__((always_inline)) {
        const Loads L = begin_loads(it);
#pragma unroll
        for (int i = 0; i < NI; i++) dma_k(L, (it + 2) % 3, i);
#pragma unroll
        for (int i = 0; i < NI; i++) dma_v(L, it & 1, i);
    };
    auto end_iteration = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the loads is in LDS
        __syncthreads();
    };
    // S(n) from the K tile at `kb`, alone (first tile of the wave)
    auto s_tile = [&](f32x16 (&s)[2], const char *kb) __attribute__((always_inline)) {
        static_for<2>([&](auto KH) {
            constexpr int kh = decltype(KH)::value;
            static_for<KS>([&](auto KSI) {
                constexpr int ks = decltype(KSI)::value;
                i32x4 kf = *(const i32x4 *)(kb + k_rd(ks) + 32 * kh * ROWB);
                settle(kf);
                mma_s<ks, BF16, ks == 0>(s[kh], kf);
            });
        });
    };
    // the vector half of phase 2 for tile n: mask, row max, first half of the exponentials
    auto start_softmax = [&](f32x16 (&s)[2], const int n) __attribute__((always_inline)) {
        if (needs_mask(n)) mask_tile(s, n);
        row_max(s);
        exp_half(s, std::integral_constant<int, 0>{});
    };

    // ---- prologue: K of the first two tiles ----
    load_k_now(n_max - 1, 0);
    if (n_it > 1) load_k_now(n_max - 2, 1);
    // ---- Q fragments (B operand of S^T = K Q^T) into a[128:191]: fragment s, lane (r, hh) holds
    // Q[q0w + r][16 s + 8 hh ..].  All sixteen requests go out behind the first K tiles' (one memory
    // round trip for the lot), then they are moved to their registers. ----
    {
        const int row = q0w + r;
        const bool rok = row < sq;
        const unsigned short *qr = qg + (int64_t)(rok ? row : 0) * p.q_rs;
        i32x4 qv[KS];
#pragma unroll
        for (int s_ = 0; s_ < KS; s_++) {
            const int col = 16 * s_ + 8 * hh;
            qv[s_] = (rok && col < p.d) ? *(const i32x4 *)(qr + col) : i32x4{0, 0, 0, 0};
        }
        static_for<KS>([&](auto S_) { q_write<decltype(S_)::value>(qv[decltype(S_)::value]); });
    }
    if constexpr (PAGED) {
        pq0 = page_of(n_max - 1);
        pq1 = page_of(n_max - 2);
        pq2 = page_of(n_max - 3);
    }
    end_iteration();

    // iterations before this wave's first tile: it only helps with the loads
    int it = 0;
    for (; it < skip; it++) {
        issue_loads(it);
        end_iteration();
    }
    if (it < n_it) {
        // ---- the wave's first tile: S and the start of its softmax, nothing to overlap with ----
        issue_loads(it);
        if (it & 1) {
            s_tile(sB, smem + (it % 3) * TILE);
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // S has landed
            start_softmax(sB, n_max - 1 - it);
        } else {
            s_tile(sA, smem + (it % 3) * TILE);
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
            start_softmax(sA, n_max - 1 - it);
        }
        rescale_pend = false;  // nothing accumulated yet
        end_iteration();
        it++;

        // One steady iteration: S(n) into sNew while the softmax of tile n + 1 (in sOld) finishes, then
        // P V of tile n + 1 while the softmax of tile n starts.  The LDS reads run RING - 1 MFMAs
        // ahead, the direct-to-LDS loads are spread over phase 1.
        // PAR = it_ & 1 picks the V slots (V of the previous tile in slot PAR ^ 1) and the S buffers
        // (S(n) into sB on odd iterations), all compile-time.
        auto steady = [&](const int it_, auto PARC) __attribute__((always_inline)) {
            constexpr int PAR = decltype(PARC)::value;
            f32x16 (&sNew)[2] = PAR ? sB : sA;
            f32x16 (&sOld)[2] = PAR ? sA : sB;
            const int n = n_max - 1 - it_;
            const int kslot = it_ % 3;
            const char *kb = smem + kslot * TILE;
            const char *vb = smem + (3 + (PAR ^ 1)) * TILE;
            const int kslot_next = kslot == 0 ? 2 : kslot - 1;  // (it_ + 2) % 3
            const Loads L = begin_loads(it_);
            // ---- phase 1: S(n) | second half of softmax(n + 1) | the loads of this iteration ----
            constexpr int RING = WIDE_RING;  // fragments read ahead of their MFMAs (+1)
            i32x4 kfr[RING];
            static_for<RING - 1>([&](auto I) { kfr[decltype(I)::value] = *(const i32x4 *)(kb + k_rd(decltype(I)::value)); });
            static_for<2 * KS>([&](auto ST) {
                constexpr int st = decltype(ST)::value, kh = st / KS, ks = st % KS;
                constexpr int nx = st + RING - 1;
                if constexpr (nx < 2 * KS)
                    kfr[nx % RING] = *(const i32x4 *)(kb + k_rd(nx % KS) + 32 * (nx / KS) * ROWB);
                mma_s<ks, BF16, ks == 0>(sNew[kh], kfr[st % RING]);
                // the eight pairs of the old tile's key half 1 at even distances over the 2 KS MFMAs
                // (beside every fourth one, from the second, at KS = 16)
                static_for<8>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (i * 2 * KS / 8 + 1 == st)
                        exp_pair(sOld, std::integral_constant<int, 1>{}, std::integral_constant<int, (i >> 2)>{},
                                 std::integral_constant<int, (i & 3)>{});
                });
                // ... and the 2 NI loads (beside every other MFMA at KS = 16)
                static_for<2 * NI>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (i * 2 * KS / (2 * NI) == st) {
                        if constexpr (i < NI) dma_k(L, kslot_next, i);
                        else dma_v(L, PAR, i - NI);
                    }
                });
                if constexpr (st + 1 < 2 * KS) ready(kfr[(st + 1) % RING]);  // (after this step's vector work)
            });
            // ---- phase 2: P V of tile n + 1 | row max and first half of softmax(n) ----
            operand_fence(pf[0], pf[1], pf[2], pf[3]);
            if (needs_mask(n)) {
                // tiles on a mask edge (few per row block): products first, then the vector work
                pv_tile(vb);
                mask_tile(sNew, n);
                row_max(sNew);
                exp_half(sNew, std::integral_constant<int, 0>{});
            } else {
                i32x4 vfr[RING];
                static_for<RING - 1>([&](auto I) { vfr[decltype(I)::value] = v_frag(vb, I); });
                float mxa = -INFINITY;
                // k-steps 0, 1 beside the row maximum of S(n): the first MFMAs keep the first read of
                // S(n) behind its last MFMA
                static_for<2 * NDB>([&](auto G) {
                    constexpr int g = decltype(G)::value;
                    vfr[(g + RING - 1) % RING] = v_frag(vb, std::integral_constant<int, g + RING - 1>{});
                    mma_o<g % NDB, BF16>(vfr[g % RING], pf[g / NDB]);
                    ready(vfr[(g + 1) % RING]);
                    if constexpr (g >= 4) {
                        // the steps from the fifth on share the 16 max3 of the tile
                        constexpr int M = 2 * NDB - 4;
                        constexpr int lo = (g - 4) * 16 / M, hi = (g - 3) * 16 / M;
#pragma unroll
                        for (int j = lo; j < hi; j++) mxa = max3(mxa, sNew[0][j], sNew[1][j]);
                    }
                });
                move_reference(half_swap_max(mxa));
                // k-steps 2, 3 beside the first half of the new tile's exponentials
                static_for<2 * NDB>([&](auto G8) {
                    constexpr int g = 2 * NDB + decltype(G8)::value;
                    if constexpr (g + RING - 1 < 4 * NDB) vfr[(g + RING - 1) % RING] = v_frag(vb, std::integral_constant<int, g + RING - 1>{});
                    mma_o<g % NDB, BF16>(vfr[g % RING], pf[g / NDB]);
                    if constexpr (g + 1 < 4 * NDB) ready(vfr[(g + 1) % RING]);
                    // the eight pairs of key half 0 at even distances over these 2 NDB MFMAs (every other one at NDB = 8)
                    static_for<8>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        if constexpr (i * 2 * NDB / 8 + 1 == g - 2 * NDB)
                            exp_pair(sNew, std::integral_constant<int, 0>{}, std::integral_constant<int, (i >> 2)>{},
                                     std::integral_constant<int, (i & 3)>{});
                    });
                });
            }
            if (rescale_pend) apply_rescale();
            end_iteration();
        };
        if (it < n_it && (it & 1)) { steady(it, std::integral_constant<int, 1>{}); it++; }
        while (it < n_it) {
            steady(it, std::integral_constant<int, 0>{});
            if (++it >= n_it) break;
            steady(it, std::integral_constant<int, 1>{});
            ++it;
        }
        // ---- drain: second half of the last tile's softmax and its P V ----
        const char *vb = smem + (3 + ((n_it - 1) & 1)) * TILE;
        if ((n_it - 1) & 1) exp_half(sB, std::integral_constant<int, 1>{});
        else exp_half(sA, std::integral_constant<int, 1>{});
        operand_fence(pf[0], pf[1], pf[2], pf[3]);
        pv_tile(vb);
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMA results have landed before they are read
    __syncthreads();  // every wave is done with the K/V tiles before the O staging overwrites them

    // ---- epilogue: normalise, LSE (softmax.h:168-185), store O through LDS as whole rows ----
    char *ob = smem + wave * QM * O_ROWB;
    const float l_tot = half_swap_sum(l_run);
    const bool bad = (l_tot == 0.f) || (l_tot != l_tot);
    const float inv = bad ? 1.f : 1.f / l_tot;
    if (hh == 0 && q0w + r < sq) lse_g[q0w + r] = bad ? INFINITY : m_run * p.scale + __logf(l_tot);
    static_for<NDB>([&](auto DB) {
        constexpr int db = decltype(DB)::value;
        static_for<4>([&](auto G4) {
            constexpr int g4 = decltype(G4)::value;
            i32x2 w;
            w[0] = pack2<BF16>(o_read<db, 4 * g4 + 0>() * inv, o_read<db, 4 * g4 + 1>() * inv);
            w[1] = pack2<BF16>(o_read<db, 4 * g4 + 2>() * inv, o_read<db, 4 * g4 + 3>() * inv);
            *(i32x2 *)(ob + r * O_ROWB + (32 * db + 8 * g4 + 4 * hh) * 2) = w;
        });
    });
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's own LDS writes
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < (QM * CPR) / 64; i++) {
        const int c = lane + 64 * i;
        const int row = c / CPR, ch = c % CPR;
        const i32x4 w = *(const i32x4 *)(ob + row * O_ROWB + ch * 16);
        if (q0w + row < sq && ch * 8 < p.d) *(i32x4 *)(og + (int64_t)(q0w + row) * p.o_rs + ch * 8) = w;
    }
}

// The launcher of one K/V layout (template parameter PAGED); the paged instantiations are built as a
// translation unit of their own (fwd_prefill_wide_paged.hip), which halves the build's longest job.
template <bool PAGED>
static hipError_t launch_wide_fwd(const KParams &p, hipStream_t stream) {
    const int n_mblocks = (p.seqlen_q + fw256::BM - 1) / fw256::BM;
    const long long blocks = (long long)n_mblocks * p.b * p.h;
    if (blocks <= 0) return hipSuccess;
    static std::atomic<uint64_t> lds_set[6];
    auto launch = [&](auto kern, std::atomic<uint64_t> &flag) -> hipError_t {
        if (hipError_t e = set_dyn_lds_once((const void *)kern, fw256::LDS_BYTES, flag); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), fw256::LDS_BYTES, stream, p, n_mblocks);
        return hipGetLastError();
    };
    // head dims up to 160 / 192 stop their products at that column (b4 s4096 h16 d192 causal 0.557 -> 0.496 ms, d160 0.544 -> 0.455)
    auto pick = [&](auto DVC, const int at) -> hipError_t {
        constexpr int DV = decltype(DVC)::value;
        return p.is_bf16 ? launch(fwd_prefill_wide_kernel<true, PAGED, DV>, lds_set[at]) : launch(fwd_prefill_wide_kernel<false, PAGED, DV>, lds_set[at + 1]);
    };
    if (p.d <= 160) return pick(std::integral_constant<int, 160>{}, 0);
    if (p.d <= 192) return pick(std::integral_constant<int, 192>{}, 2);
    return pick(std::integral_constant<int, 256>{}, 4);
}

#ifdef VFA_WIDE_PAGED_TU
hipError_t run_fwd_prefill_wide_paged(const KParams &p, hipStream_t stream) { return launch_wide_fwd<true>(p, stream); }
#else
hipError_t run_fwd_prefill_wide_paged(const KParams &p, hipStream_t stream);   // fwd_prefill_wide_paged.hip

// Whether this kernel takes the call: head dims 129..256, K/V contiguous per (batch, head) within a
// 32-bit buffer offset, no ALiBi (the caller checks dropout and split).
bool fwd_prefill_wide_ok(const KParams &p) {
    if (p.d <= 128 || p.d > 256 || p.alibi_slopes) return false;
    const int64_t lim = (int64_t)1 << 30;
    if (p.block_table)   // pages: a power of two, whole 16-row groups, 32-bit offsets inside a page
        return p.page_shift >= 0 && p.page_size % 16 == 0 && (int64_t)p.page_size * p.k_rs * 2 < lim &&
               (int64_t)p.page_size * p.v_rs * 2 < lim;
    const int64_t rows = (int64_t)p.seqlen_k + 256;
    return rows * p.k_rs * 2 < lim && rows * p.v_rs * 2 < lim;
}

hipError_t run_fwd_prefill_wide(const KParams &p, hipStream_t stream) {
    return p.block_table ? run_fwd_prefill_wide_paged(p, stream) : launch_wide_fwd<false>(p, stream);
}
#endif

}  // namespace vfa
