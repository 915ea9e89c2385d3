// Forward kernel with 64 query rows per wave at ONE wave per SIMD (head dim up to 128, contiguous
// K/V, no dropout): the same tile loop as fwd_prefill_kernel (flash_fwd_kernel.h:27-470), re-cut
// for the LDS bandwidth of a CU.
//
// In the 8-wave kernel every wave reads the whole 64-key K and V tile from LDS (32 KiB) for its 32
// rows: 256 KiB per tile and CU, half of what the LDS array can deliver in the 2048 cycles the
// tile's MFMAs take on a SIMD.
// Here a wave owns two 32-row halves and every K fragment and every transposed V fragment it reads
// feeds two MFMAs: half the LDS traffic per MFMA.  With one wave per SIMD nothing but the wave's own
// instruction stream hides its vector work, so the tile loop is software-pipelined by hand:
//
//   iteration i:  phase 1   S(i) = K(i) Q^T            (32 MFMAs)  | second half of softmax(i-1)
//                 phase 2   O += V(i-1)^T P(i-1)^T     (32 MFMAs)  | mask / row max of S(i), first half of softmax(i)
//
// S is double-buffered in the arch VGPRs; O (128 registers) and the Q fragments (64) live in the
// accumulator half of the register file at FIXED registers (a[0:127], a[128:191]) that only the asm
// statements of fwd_w64_regs.inc touch: values the compiler allocates there are copied between the
// two files at every branch join, which costs more than the vector work of a tile.  The MFMAs are
// therefore opaque to the compiler; the wait states between them and the vector instructions that
// feed or read them are placed by hand (operand_fence, the s_nop groups).  K(i+1) and V(i) are fetched straight
// into LDS (asm-issued, so that the compiler's wait-count pass does not hold LDS reads behind them)
// at the top of iteration i, one barrier per iteration.
//
// This translation unit is compiled with -mllvm -amdgpu-mfma-vgpr-form=1 (MFMA results of the S
// products in VGPRs: by default a kernel that may use more than 256 registers gets them in AGPRs).
#include "vfa_kernel_params.h"
#include "vfa_one_wave_helpers.h"
#include <type_traits>

namespace vfa {

#ifndef W64_RING
#define W64_RING 3
#endif

namespace f64 {
constexpr int DP = 128, BN = 64, NW = 4, QM = 64, BM = NW * QM;
constexpr int CPR = DP / 8, ROWB = DP * 2, TILE = BN * ROWB;
constexpr int KS = DP / 16, NDB = DP / 32;
constexpr int O_ROWB = DP * 2 + 16;
constexpr int KV_LDS = 5 * TILE;                 // K0 K1 K2 V0 V1
constexpr int O_LDS = NW * QM * O_ROWB;
constexpr int LDS_BYTES = KV_LDS > O_LDS ? KV_LDS : O_LDS;
}  // namespace f64

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
__device__ __forceinline__ void ready(i32x4 &x) { asm volatile("" : "+v"(x)); }
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

template <bool BF16>
__global__ void __launch_bounds__(256, 1)
fwd_prefill_w64_kernel(const KParams p, const int n_mblocks) {
    using namespace f64;
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
    const unsigned short *kg = (const unsigned short *)p.k + (int64_t)hk * p.k_hs + bi.k_off(p.k_bs, p.k_rs, bidb_cache);
    const unsigned short *vg = (const unsigned short *)p.v + (int64_t)hk * p.v_hs + bi.k_off(p.v_bs, p.v_rs, bidb_cache);
    const i32x4 k_rsrc = make_rsrc(kg, (unsigned)(((int64_t)(sk - 1) * p.k_rs + p.d) * 2));
    const i32x4 v_rsrc = make_rsrc(vg, (unsigned)(((int64_t)(sk - 1) * p.v_rs + p.d) * 2));
    const i32x4 k_rsrc_none = make_rsrc(kg, 0u);  // every lane out of range: the load writes zeros
    // a wave instruction fills 1 KiB of LDS in lane order = 4 rows of the tile; lane -> the chunk the
    // XOR swizzle places at its slot (rows 16 apart share the swizzle); out-of-range lanes write zeros
    constexpr int RPI = 64 / CPR, NI = BN / (RPI * NW);
    unsigned dk_voff, dv_voff;
    {
        const int row0 = RPI * wave + lane / CPR;
        const int ch = (lane % CPR) ^ swz<DP>(row0);
        const bool ok = ch * 8 < p.d;
        dk_voff = ok ? (unsigned)((row0 * (int)p.k_rs + ch * 8) * 2) : 0x80000000u;
        dv_voff = ok ? (unsigned)((row0 * (int)p.v_rs + ch * 8) * 2) : 0x80000000u;
    }
    const unsigned k_rs2 = (unsigned)p.k_rs * 2, v_rs2 = (unsigned)p.v_rs * 2;
    const unsigned lds0 = 0;  // the dynamic LDS segment starts at LDS address 0 (no static LDS here)
    auto dma_k = [&](const int n, const int slot, const int i, const bool exists = true) __attribute__((always_inline)) {
        const unsigned rows = (unsigned)(n * BN + RPI * NW * i);
        i32x4 rs;  // (a tile that does not exist is fetched through the empty descriptor: no branch in the loop)
#pragma unroll
        for (int t = 0; t < 4; t++) rs[t] = exists ? k_rsrc[t] : k_rsrc_none[t];
        dma_16b_to_lds(rs, dk_voff + rows * k_rs2, lds0 + slot * TILE + (wave + NW * i) * 1024);
    };
    auto dma_v = [&](const int n, const int slot, const int i) __attribute__((always_inline)) {
        const unsigned rows = (unsigned)(n * BN + RPI * NW * i);
        dma_16b_to_lds(v_rsrc, dv_voff + rows * v_rs2, lds0 + (3 + slot) * TILE + (wave + NW * i) * 1024);
    };

    // ---- Q fragments (B operand of S^T = K Q^T) into a[128:191]: fragment 8 qh + s, lane (r, hh)
    // holds Q[q0w + 32 qh + r][16 s + 8 hh ..] ----
    static_for<2>([&](auto QH) {
        constexpr int qh = decltype(QH)::value;
        const int row = q0w + 32 * qh + r;
        const bool rok = row < sq;
        const unsigned short *qr = qg + (int64_t)(rok ? row : 0) * p.q_rs;
        static_for<KS>([&](auto S_) {
            constexpr int s_ = decltype(S_)::value;
            const int col = 16 * s_ + 8 * hh;
            const i32x4 v = (rok && col < p.d) ? *(const i32x4 *)(qr + col) : i32x4{0, 0, 0, 0};
            q_write<8 * qh + s_>(v);
        });
    });
    static_for<8>([&](auto T) { o_zero<decltype(T)::value>(); });  // O^T tile 4 qh + db
    float m_run[2] = {-INFINITY, -INFINITY};  // reference point of the exponentials, followed lazily
    float l_run[2] = {0.f, 0.f};              // running row sums, partial per lane half
    const float c2 = p.scale_log2;

    // per-lane LDS read offsets
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
    int k_rd[KS], v_rd[NDB][2];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) k_rd[ks] = lds_off<DP>(r, 2 * ks + hh);
#pragma unroll
    for (int db = 0; db < NDB; db++)
#pragma unroll
        for (int e = 0; e < 2; e++)
            v_rd[db][e] = lds_off<DP>(4 * hh + tr_q + 8 * e, 4 * db + 2 * tr_g + (tr_p >> 1)) + 8 * (tr_p & 1);

    // Tiles this wave's rows see, a sub-range of the block's: the block's first iterations cover
    // tiles above the diagonal of the lower waves' rows; those waves only help with the loads.
    int n_top = n_max;
    if (p.is_causal || p.is_local) n_top = min(n_max, (q0w + QM + sk - sq + p.wr + BN - 1) / BN);
    if (q0w >= sq) n_top = n_min;  // a wave beyond the last row
    n_top = max(n_top, n_min);
    const int n_it = n_max - n_min;           // block iterations (tile n_max - 1 - it)
    const int skip = n_max - n_top;           // leading iterations without work for this wave

    f32x16 sA[2][2], sB[2][2];  // S^T tiles [query half][key half], two pipeline stages
    i32x4 pf[2][4];             // P^T fragments [query half][k-step of 16 keys] of the tile in flight
    float mc[2] = {0.f, 0.f};   // reference points times scale_log2 (of the tile whose P is being built)
    float nmc[2] = {0.f, 0.f};  // their negatives (the addend of the exponent's multiply-add)
    float alpha_pend[2] = {1.f, 1.f};
    bool rescale_pend = false;

    // ---- pieces of one tile's work ----
    // masks of tile n (wave-uniform test outside): causal / local / key tail (mask.h:128-209)
    auto mask_tile = [&](f32x16 (&s)[2][2], const int n) __attribute__((always_inline)) {
        const int c0 = n * BN;
#pragma unroll
        for (int qh = 0; qh < 2; qh++)
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int j = 0; j < 16; j++) asm volatile("" : "+v"(s[qh][kh][j]));  // reads stay behind the MFMAs before this point
#pragma unroll
        for (int qh = 0; qh < 2; qh++) {
            const int row = q0w + 32 * qh + r;
            const int lo = p.is_local ? max(0, row + sk - sq - p.wl) : 0;
            const int hi = (p.is_causal || p.is_local) ? min(sk, row + 1 + sk - sq + p.wr) : sk;
            // register j of key half kh is key c0 + 32 kh + (j & 3) + 8 (j >> 2) + 4 hh
            const int lo_r = lo - c0 - 4 * hh, hi_r = hi - c0 - 4 * hh;
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int off = 32 * kh + (j & 3) + 8 * (j >> 2);
                    s[qh][kh][j] = (off >= lo_r && off < hi_r) ? s[qh][kh][j] : -INFINITY;
                }
        }
    };
    auto needs_mask = [&](const int n) {
        const int c0 = n * BN;
        bool need = c0 + BN > sk;
        if (p.is_causal || p.is_local) need = need || (c0 + BN > q0w + 1 + sk - sq + p.wr);
        if (p.is_local) need = need || (c0 < q0w + QM - 1 + sk - sq - p.wl);
        return need;
    };
    // row maxima of a tile and the (lazy) move of the reference point: softmax.h:135-166
#ifndef W64_RESCALE
#define W64_RESCALE 8.f
#endif
    constexpr float RESCALE_LOG2 = W64_RESCALE;
    auto row_max = [&](f32x16 (&s)[2][2]) __attribute__((always_inline)) {
        float mx[2];
#pragma unroll
        for (int qh = 0; qh < 2; qh++) {
            float a = -INFINITY;
#pragma unroll
            for (int j = 0; j < 16; j++) a = max3(a, s[qh][0][j], s[qh][1][j]);
            mx[qh] = half_swap_max(a);
        }
        const bool upd = __any(((mx[0] - m_run[0]) * c2 > RESCALE_LOG2) || ((mx[1] - m_run[1]) * c2 > RESCALE_LOG2));
        if (upd) {
#pragma unroll
            for (int qh = 0; qh < 2; qh++) {
                const float m_new = fmaxf(m_run[qh], mx[qh]);
                const float m_cur = (m_new == -INFINITY) ? 0.f : m_new;
                const float al = __builtin_amdgcn_exp2f((m_run[qh] - m_cur) * c2);
                l_run[qh] *= al;
                alpha_pend[qh] = al;   // O is rescaled once the P V products in flight have landed
                m_run[qh] = m_new;
            }
            rescale_pend = true;
        }
#pragma unroll
        for (int qh = 0; qh < 2; qh++) {
            mc[qh] = ((m_run[qh] == -INFINITY) ? 0.f : m_run[qh]) * c2;
            nmc[qh] = -mc[qh];
        }
    };
    // exponentials of 16 keys (k-step e2 of key half kh) of a tile -> one P fragment, row sums
    auto exp_step = [&](f32x16 (&s)[2][2], const int kh, const int qh, const int e2) __attribute__((always_inline)) {
        float ps = 0.f;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int j = 8 * e2 + 2 * w;
            float x0, x1;  // (volatile first reads of the scores: see exp_pair)
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x0) : "v"(s[qh][kh][j]), "s"(c2), "v"(nmc[qh]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x1) : "v"(s[qh][kh][j + 1]), "s"(c2), "v"(nmc[qh]));
            const float e0 = __builtin_amdgcn_exp2f(x0);
            const float e1 = __builtin_amdgcn_exp2f(x1);
            ps += e0 + e1;
            pf[qh][2 * kh + e2][w] = pack2<BF16>(e0, e1);
        }
        l_run[qh] += ps;
    };
    // O^T += V^T P^T for k-step KSI (16 keys) of the V tile at `vb` (both query halves per V fragment)
    auto pv_step = [&](const char *vb, auto KSI) __attribute__((always_inline)) {
        constexpr int ks = decltype(KSI)::value;
        static_for<NDB>([&](auto DB) {
            constexpr int db = decltype(DB)::value;
            const i32x2 a0 = __builtin_bit_cast(i32x2, lds_read_tr16(vb + v_rd[db][0] + 16 * ks * ROWB));
            const i32x2 a1 = __builtin_bit_cast(i32x2, lds_read_tr16(vb + v_rd[db][1] + 16 * ks * ROWB));
            i32x4 vf = {a0[0], a0[1], a1[0], a1[1]};
            settle(vf);
            mma_o<db, BF16>(vf, pf[0][ks]);
            mma_o<4 + db, BF16>(vf, pf[1][ks]);
        });
    };
    auto apply_rescale = [&]() __attribute__((always_inline)) {
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // the last P V results have landed
        static_for<8>([&](auto T) { o_scale<decltype(T)::value>(alpha_pend[decltype(T)::value / 4]); });
        asm volatile("s_nop 3" ::: "memory");
        rescale_pend = false;
    };
    // K tiles go through a ring of three LDS slots (tile of iteration it in slot it % 3, fetched two
    // iterations ahead), V tiles through two (fetched in the iteration of their S, used by the P V of
    // the next one).
    auto issue_loads = [&](const int it) __attribute__((always_inline)) {
        const int n = n_max - 1 - it;
        if (n - 2 >= n_min) {
#pragma unroll
            for (int i = 0; i < NI; i++) dma_k(n - 2, (it + 2) % 3, i);
        }
#pragma unroll
        for (int i = 0; i < NI; i++) dma_v(n, it & 1, i);
    };
    auto end_iteration = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the loads is in LDS
        __syncthreads();
    };
    // S(n) from the K tile at `kb`, alone (first tile of the wave)
    auto s_tile = [&](f32x16 (&s)[2][2], const char *kb) __attribute__((always_inline)) {
        static_for<2>([&](auto KH) {
            constexpr int kh = decltype(KH)::value;
            static_for<KS>([&](auto KSI) {
                constexpr int ks = decltype(KSI)::value;
                i32x4 kf = *(const i32x4 *)(kb + k_rd[ks] + 32 * kh * ROWB);
                settle(kf);
                mma_s<ks, BF16, ks == 0>(s[0][kh], kf);
                mma_s<8 + ks, BF16, ks == 0>(s[1][kh], kf);
            });
        });
    };
    // the vector half of phase 2 for tile n: mask, row max, first half of the exponentials
    auto start_softmax = [&](f32x16 (&s)[2][2], const int n) __attribute__((always_inline)) {
        if (needs_mask(n)) mask_tile(s, n);
        row_max(s);
#pragma unroll
        for (int qh = 0; qh < 2; qh++) {
            exp_step(s, 0, qh, 0);
            exp_step(s, 0, qh, 1);
        }
    };

    // ---- prologue: K of the first two tiles ----
#pragma unroll
    for (int i = 0; i < NI; i++) dma_k(n_max - 1, 0, i);
    if (n_it > 1) {
#pragma unroll
        for (int i = 0; i < NI; i++) dma_k(n_max - 2, 1, i);
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

        // two exponentials of the tile in `s_`: P dword W of k-step E2 of key half KH, query half QH
        auto exp_pair = [&](f32x16 (&s_)[2][2], auto KH, auto QH, auto E2, auto W) __attribute__((always_inline)) {
            constexpr int kh = decltype(KH)::value, qh = decltype(QH)::value, e2 = decltype(E2)::value, w = decltype(W)::value;
            constexpr int j = 8 * e2 + 2 * w;
            // The scores come out of asm MFMAs: their first read is a volatile statement, which keeps
            // its place among the MFMA statements (the compiler would otherwise move the read up to
            // right behind the MFMA that produced the score, before the result has landed).
            float x0, x1;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x0) : "v"(s_[qh][kh][j]), "s"(c2), "v"(nmc[qh]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x1) : "v"(s_[qh][kh][j + 1]), "s"(c2), "v"(nmc[qh]));
            const float e0 = __builtin_amdgcn_exp2f(x0);
            const float e1 = __builtin_amdgcn_exp2f(x1);
            l_run[qh] += e0 + e1;
            pf[qh][2 * kh + e2][w] = pack2<BF16>(e0, e1);
        };
        // V fragment (transposed read) G = 4 ks + db of the V tile at `vb`
        auto v_frag = [&](const char *vb, auto G) __attribute__((always_inline)) -> i32x4 {
            constexpr int ks = decltype(G)::value / NDB, db = decltype(G)::value % NDB;
            const i32x2 a0 = __builtin_bit_cast(i32x2, lds_read_tr16(vb + v_rd[db][0] + 16 * ks * ROWB));
            const i32x2 a1 = __builtin_bit_cast(i32x2, lds_read_tr16(vb + v_rd[db][1] + 16 * ks * ROWB));
            return i32x4{a0[0], a0[1], a1[0], a1[1]};
        };

        // One steady iteration: S(n) into sNew while the softmax of tile n + 1 (in sOld) finishes, then
        // P V of tile n + 1 while the softmax of tile n starts.  Every MFMA pair carries its share of
        // the vector work, the LDS reads run two steps ahead, the direct-to-LDS loads are spread over
        // phase 1.
        // PAR = it_ & 1 picks the LDS slots (K of this tile in slot PAR, V of the previous tile in
        // slot PAR ^ 1) and the S buffers (S(n) into sB on odd iterations), all compile-time.
        auto steady = [&](const int it_, auto PARC) __attribute__((always_inline)) {
            constexpr int PAR = decltype(PARC)::value;
            f32x16 (&sNew)[2][2] = PAR ? sB : sA;
            f32x16 (&sOld)[2][2] = PAR ? sA : sB;
            const int n = n_max - 1 - it_;
            const int kslot = it_ % 3;
            const char *kb = smem + kslot * TILE;
            const char *vb = smem + (3 + (PAR ^ 1)) * TILE;
            const bool k_next = n - 2 >= n_min;
            const int kslot_next = kslot == 0 ? 2 : kslot - 1;  // (it_ + 2) % 3
            // ---- phase 1: S(n) | second half of softmax(n + 1) | the loads of this iteration ----
            constexpr int RING = W64_RING;  // fragments read ahead of their MFMAs (+1)
            i32x4 kfr[RING];
            static_for<RING - 1>([&](auto I) { kfr[decltype(I)::value] = *(const i32x4 *)(kb + k_rd[decltype(I)::value]); });
            static_for<16>([&](auto ST) {
                constexpr int st = decltype(ST)::value, kh = st >> 3, ks = st & 7;
                constexpr int nx = st + RING - 1;
                if constexpr (nx < 16)
                    kfr[nx % RING] = *(const i32x4 *)(kb + k_rd[nx & 7] + 32 * (nx >> 3) * ROWB);
                mma_s<ks, BF16, ks == 0>(sNew[0][kh], kfr[st % RING]);
                mma_s<8 + ks, BF16, ks == 0>(sNew[1][kh], kfr[st % RING]);
                // P dword st of the old tile's key half 1: query half st >> 3, k-step (st >> 2) & 1, dword st & 3
                exp_pair(sOld, std::integral_constant<int, 1>{}, std::integral_constant<int, (st >> 3)>{},
                         std::integral_constant<int, ((st >> 2) & 1)>{}, std::integral_constant<int, (st & 3)>{});
                if constexpr ((st & 1) == 1) {
                    constexpr int i = st >> 1;
                    if constexpr (i < NI) dma_k(n - 2, kslot_next, i, k_next);
                    else dma_v(n, PAR, i - NI);
                }
                if constexpr (st + 1 < 16) ready(kfr[(st + 1) % RING]);  // (after this step's vector work)
            });
            // ---- phase 2: P V of tile n + 1 | row max and first half of softmax(n) ----
            operand_fence(pf[0][0], pf[1][0], pf[0][1], pf[1][1]);
            operand_fence(pf[0][2], pf[1][2], pf[0][3], pf[1][3]);
            if (needs_mask(n)) {
                // tiles on a mask edge (few per row block): products first, then the vector work
                static_for<16>([&](auto G) {
                    constexpr int g = decltype(G)::value;
                    i32x4 vf = v_frag(vb, G);
                    settle(vf);
                    mma_o<g % NDB, BF16>(vf, pf[0][g / NDB]);
                    mma_o<4 + g % NDB, BF16>(vf, pf[1][g / NDB]);
                });
                mask_tile(sNew, n);
                row_max(sNew);
                static_for<16>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    exp_pair(sNew, std::integral_constant<int, 0>{}, std::integral_constant<int, (i >> 3)>{},
                             std::integral_constant<int, ((i >> 2) & 1)>{}, std::integral_constant<int, (i & 3)>{});
                });
            } else {
                i32x4 vfr[RING];
                static_for<RING - 1>([&](auto I) { vfr[decltype(I)::value] = v_frag(vb, I); });
                float mxa[2] = {-INFINITY, -INFINITY};
                // k-steps 0, 1 (the 16 keys x 2 of key half 0) beside the row maxima of S(n): the first
                // MFMA pairs keep the first read of S(n) behind its last MFMA
                static_for<8>([&](auto G) {
                    constexpr int g = decltype(G)::value;
                    vfr[(g + RING - 1) % RING] = v_frag(vb, std::integral_constant<int, g + RING - 1>{});
                    mma_o<g % NDB, BF16>(vfr[g % RING], pf[0][g / NDB]);
                    mma_o<4 + g % NDB, BF16>(vfr[g % RING], pf[1][g / NDB]);
                    ready(vfr[(g + 1) % RING]);
                    if constexpr (g >= 2) {
                        // 6 steps x (2 query halves x ... ): values 16 (g - 2) / 6 ...
                        constexpr int lo = (g - 2) * 16 / 6, hi = (g - 1) * 16 / 6;
#pragma unroll
                        for (int qh = 0; qh < 2; qh++)
#pragma unroll
                            for (int j = lo; j < hi; j++) mxa[qh] = max3(mxa[qh], sNew[qh][0][j], sNew[qh][1][j]);
                    }
                });
                {
                    float mx[2] = {half_swap_max(mxa[0]), half_swap_max(mxa[1])};
                    const bool upd = __any(((mx[0] - m_run[0]) * c2 > RESCALE_LOG2) || ((mx[1] - m_run[1]) * c2 > RESCALE_LOG2));
                    if (upd) {
#pragma unroll
                        for (int qh = 0; qh < 2; qh++) {
                            const float m_new = fmaxf(m_run[qh], mx[qh]);
                            const float m_cur = (m_new == -INFINITY) ? 0.f : m_new;
                            const float al = __builtin_amdgcn_exp2f((m_run[qh] - m_cur) * c2);
                            l_run[qh] *= al;
                            alpha_pend[qh] = al;
                            m_run[qh] = m_new;
                            mc[qh] = m_cur * c2;
                            nmc[qh] = -mc[qh];
                        }
                        rescale_pend = true;
                    }
                }
                // k-steps 2, 3 beside the first half of the new tile's exponentials
                static_for<8>([&](auto G8) {
                    constexpr int g = 8 + decltype(G8)::value;
                    if constexpr (g + RING - 1 < 16) vfr[(g + RING - 1) % RING] = v_frag(vb, std::integral_constant<int, g + RING - 1>{});
                    mma_o<g % NDB, BF16>(vfr[g % RING], pf[0][g / NDB]);
                    mma_o<4 + g % NDB, BF16>(vfr[g % RING], pf[1][g / NDB]);
                    if constexpr (g + 1 < 16) ready(vfr[(g + 1) % RING]);
                    constexpr int i0 = 2 * (g - 8);
                    exp_pair(sNew, std::integral_constant<int, 0>{}, std::integral_constant<int, (i0 >> 3)>{},
                             std::integral_constant<int, ((i0 >> 2) & 1)>{}, std::integral_constant<int, (i0 & 3)>{});
                    exp_pair(sNew, std::integral_constant<int, 0>{}, std::integral_constant<int, ((i0 + 1) >> 3)>{},
                             std::integral_constant<int, (((i0 + 1) >> 2) & 1)>{}, std::integral_constant<int, ((i0 + 1) & 3)>{});
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
        if ((n_it - 1) & 1) {
#pragma unroll
            for (int qh = 0; qh < 2; qh++) { exp_step(sB, 1, qh, 0); exp_step(sB, 1, qh, 1); }
        } else {
#pragma unroll
            for (int qh = 0; qh < 2; qh++) { exp_step(sA, 1, qh, 0); exp_step(sA, 1, qh, 1); }
        }
        operand_fence(pf[0][0], pf[1][0], pf[0][1], pf[1][1]);
        operand_fence(pf[0][2], pf[1][2], pf[0][3], pf[1][3]);
        static_for<4>([&](auto KSI) { pv_step(vb, KSI); });
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMA results have landed before they are read
    __syncthreads();  // every wave is done with the K/V tiles before the O staging overwrites them

    // ---- epilogue: normalise, LSE (softmax.h:168-185), store O through LDS as whole rows ----
    char *ob = smem + wave * QM * O_ROWB;
    float inv_l[2];
#pragma unroll
    for (int qh = 0; qh < 2; qh++) {
        const float l_tot = half_swap_sum(l_run[qh]);
        const bool bad = (l_tot == 0.f) || (l_tot != l_tot);
        inv_l[qh] = bad ? 1.f : 1.f / l_tot;
        const int row = q0w + 32 * qh + r;
        if (hh == 0 && row < sq) lse_g[row] = bad ? INFINITY : m_run[qh] * p.scale + __logf(l_tot);
    }
    static_for<2>([&](auto QH) {
        constexpr int qh = decltype(QH)::value;
        static_for<NDB>([&](auto DB) {
            constexpr int db = decltype(DB)::value;
            static_for<4>([&](auto G4) {
                constexpr int g4 = decltype(G4)::value;
                const float inv = inv_l[qh];
                i32x2 w;
                w[0] = pack2<BF16>(o_read<4 * qh + db, 4 * g4 + 0>() * inv, o_read<4 * qh + db, 4 * g4 + 1>() * inv);
                w[1] = pack2<BF16>(o_read<4 * qh + db, 4 * g4 + 2>() * inv, o_read<4 * qh + db, 4 * g4 + 3>() * inv);
                *(i32x2 *)(ob + (32 * qh + r) * O_ROWB + (32 * db + 8 * g4 + 4 * hh) * 2) = w;
            });
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

// Whether this kernel takes the call: head dims 65..128, K/V contiguous per (batch, head) within a
// 32-bit buffer offset, no dropout / ALiBi / split (the caller checks dropout and split).
bool fwd_prefill_w64_ok(const KParams &p) {
    if (p.d <= 64 || p.d > 128 || p.block_table || p.alibi_slopes) return false;
    const int64_t lim = (int64_t)1 << 30;
    const int64_t rows = (int64_t)p.seqlen_k + 256;
    return rows * p.k_rs * 2 < lim && rows * p.v_rs * 2 < lim;
}

hipError_t run_fwd_prefill_w64(const KParams &p, hipStream_t stream) {
    const int n_mblocks = (p.seqlen_q + f64::BM - 1) / f64::BM;
    const long long blocks = (long long)n_mblocks * p.b * p.h;
    if (blocks <= 0) return hipSuccess;
    static std::atomic<uint64_t> lds_set_bf{0}, lds_set_fp{0};
    if (p.is_bf16) {
        auto kern = fwd_prefill_w64_kernel<true>;
        if (hipError_t e = set_dyn_lds_once((const void *)kern, f64::LDS_BYTES, lds_set_bf); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), f64::LDS_BYTES, stream, p, n_mblocks);
    } else {
        auto kern = fwd_prefill_w64_kernel<false>;
        if (hipError_t e = set_dyn_lds_once((const void *)kern, f64::LDS_BYTES, lds_set_fp); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), f64::LDS_BYTES, stream, p, n_mblocks);
    }
    return hipGetLastError();
}

}  // namespace vfa
