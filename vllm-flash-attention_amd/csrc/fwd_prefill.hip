// FlashAttention-2 forward ("prefill") for gfx950 / MI355X.
//
// Replaces compute_attn_1rowblock / flash_fwd_kernel of the reference
// (csrc/flash_attn/src/flash_fwd_kernel.h:27-470, flash_fwd_launch_template.h:29-94).
//
// Structure (wave64, MFMA 32x32x16):
//   * a workgroup of NW waves owns BM = 32*NW query rows of one (batch, head); each wave owns 32.
//   * K/V tiles of BN = 64 keys are staged HBM -> VGPR -> LDS (double buffered, one barrier per
//     tile); global loads of tile t+1 are issued before the MFMAs of tile t.
//   * S^T = K Q^T ("swapped" product): the 32x32 accumulator has the query on the lane and the
//     keys in registers, so row max / row sum are in-lane reductions plus one half-wave exchange,
//     the running max / sum / rescale factor are one scalar per lane, and P^T (converted to
//     16 bit) is directly the B operand of O^T += V^T P^T with no LDS round trip.
//   * V^T fragments come from ds_read_b64_tr_b16 (hardware transpose) on the row-major V tile.
//   * masks (causal / sliding window / key tail) and ALiBi are applied in registers, only on
//     tiles that need them (wave-uniform test).
#include "vfa_kernel_params.h"
#include "vfa_dispatch.h"
#include <type_traits>

namespace vfa {

#ifdef MFMA16_PROBE
// Timing probe only (results are wrong): the work of one 32x32x16 MFMA issued as two 16x16x32 MFMAs
// on the same operand registers -- same cycles, operand reads and accumulator registers, so that
// the clock the chip holds on the other MFMA shape can be measured before a kernel is rewritten.
typedef float f32x4p __attribute__((ext_vector_type(4)));
template <bool BF16>
__device__ __forceinline__ f32x16 mma_probe(i32x4 a, i32x4 b0, i32x4 b1, f32x16 c, const int par) {
    f32x4p c0, c1;
#pragma unroll
    for (int t = 0; t < 4; t++) { c0[t] = c[4 * par + t]; c1[t] = c[8 + 4 * par + t]; }
    if constexpr (BF16) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b0), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b1), c1, 0, 0, 0);
    } else {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b0), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b1), c1, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) { c[4 * par + t] = c0[t]; c[8 + 4 * par + t] = c1[t]; }
    return c;
}
#define MMA_S(kf, ks, acc) mma_probe<BF16>(kf, qf[(ks) & ~1], qf[(ks) | 1], (ks) < 2 ? zero : acc, (ks) & 1)
#define MMA_O(vf, ks, acc) mma_probe<BF16>(vf, pf[(ks) & ~1], pf[(ks) | 1], acc, (ks) & 1)
#else
#define MMA_S(kf, ks, acc) mma32<BF16>(kf, qf[ks], (ks) == 0 ? zero : acc)
#define MMA_O(vf, ks, acc) mma32<BF16>(vf, pf[ks], acc)
#endif


#if defined(VFA_STAMP) && defined(VFA_DROPOUT_TU)
#undef VFA_STAMP
#endif
#ifdef VFA_STAMP
// Diagnostic build only: per-phase shader-clock sums of the tile loop (never in the product build).
__device__ unsigned long long g_stamps[4096 * 8];
#define STAMP(slot)                                                        \
    do {                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                 \
        unsigned long long t_;                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                 \
        acc_[slot] += t_ - last_;                                          \
        last_ = t_;                                                        \
    } while (0)
// ... and the wall-clock (100 MHz) times at which a workgroup enters the kernel (0), has its block
// mapped (4), has requested Q and the first tile (5), has them (6), starts (1) and ends (2) its tile
// loop and leaves (3).
__device__ unsigned long long g_blk_fwd[4096 * 8];
#define FWD_BLK_STAMP(i)                                                                          \
    do {                                                                                          \
        if (wave == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {                                  \
            unsigned long long t_;                                                                \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            if (lane == 0) g_blk_fwd[blockIdx.x * 8 + (i)] = t_;                                  \
        }                                                                                         \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#define FWD_BLK_STAMP(i) do { } while (0)
#endif

constexpr int BN = 64;

template <int DP, bool BF16, int NW>
struct PrefillCfg {
    static constexpr int NT = NW * 64;
    static constexpr int BM = NW * 32;
    static constexpr int CPR = DP / 8;                    // 16-byte chunks per row
    static constexpr int TILE_CHUNKS = BN * CPR;
    static constexpr int CH = TILE_CHUNKS / NT;            // chunks per thread per tile
    static constexpr int TILE_BYTES = BN * DP * 2;
    static constexpr int O_ROWB = DP * 2 + 16;             // padded O staging row
    static constexpr int KV_LDS = 4 * TILE_BYTES;          // K0 K1 V0 V1
    static constexpr int O_LDS = NW * 32 * O_ROWB;
    // Prologue of the 8-wave kernel: the 32 Q rows of a wave pass through a private LDS image
    // (half a tile buffer) on their way to its fragments: waves 0-1 in the second K buffer, 2-3 in
    // the second V buffer (both idle until the loop), waves 4-7 behind the tile buffers.
    static constexpr int QIMG_BYTES = 32 * DP * 2;
    static constexpr int Q_LDS = KV_LDS + (NW > 4 ? (NW - 4) * QIMG_BYTES : 0);
    static constexpr int LDS0 = KV_LDS > O_LDS ? KV_LDS : O_LDS;
    static constexpr int LDS_BYTES = LDS0 > Q_LDS ? LDS0 : Q_LDS;
    static_assert(TILE_BYTES == 2 * QIMG_BYTES, "two Q images per tile buffer");
    static_assert(TILE_CHUNKS % NT == 0, "tile must divide over threads");
};

// LD = 1: K/V are contiguous per (batch, head) (no block table) and fit a 32-bit byte offset, so
// tiles are fetched with buffer loads whose descriptor ends at the last valid key row: the key
// tail and the head-dim padding read as zeros without any predication or 64-bit address
// arithmetic in the tile loop.
// LD = 2: paged K/V whose page size is a power of two and a multiple of the NT / CPR rows that
// the workgroup's threads cover per load round: the rows of one round then lie in ONE page, its
// id is a scalar, and the round is a buffer load through a descriptor built (in scalar registers)
// at that page's rows and ending at the page's or the sequence's last row.
// LD = 4: paged K/V whose page size is a power of two (and a multiple of 16 rows, as the API
// requires), written to LDS directly like LD = 3: each wave fills BN / NW consecutive rows of the
// tile, in groups of at most 16 rows that lie in one page and share a descriptor.  The variant the
// launcher takes for paged K/V (LD = 2 remains behind a development switch).
// LD = 3: as LD = 1, but the buffer loads write LDS directly (buffer_load_dwordx4 ... lds): a wave
// instruction fills 1 KiB of LDS in lane order, so each lane fetches the global chunk that the
// XOR swizzle places at its slot.  No staging registers, no ds_write, and the wave does not wait
// for its own loads before the tile's last barrier.  (Out-of-range lanes write zeros.)
// LD = 0: generic per-row addressing (any strides, any page size).
//
// DROP: dropout on P (flash_fwd_kernel.h:324-337, dropout.h) with the reference's Philox pattern,
// and the optional Return_softmax output.  The reference point of the exponentials then follows
// the running maximum exactly (tile by tile, like the reference), because S_dmask exposes it.
//
// SPLIT: blockIdx.y cuts the key range into p.num_splits parts (flash_fwd_kernel.h:504-512); every
// part writes its normalised fp32 O and its LSE (-inf if it saw no key, :513-552) to p.oaccum /
// p.lseaccum and fwd_combine_kernel merges them.  For launches whose (batch, head, row-block)
// count alone would leave most CUs idle: a few hundred query rows against a long context.
//
// PACK: the query heads that share a KV head are packed into the rows of one workgroup (row i of a
// (batch, KV head) pair = query position i / ratio, head i % ratio of the group; ratio =
// p.pack_gqa, and the launcher hands over p.h = p.h_k, p.h_ratio = 1, so the grid walks the KV
// heads).  For a few query tokens per sequence against a long cache (chunked prefill, draft-token
// verification): 16 tokens of 4 grouped heads fill half a 128-row block and sweep the keys once,
// where one block per query head sweeps them four times at an eighth of the rows each.
template <int DP, bool BF16, int NW, int LD, bool DROP, bool SPLIT = false, bool PACK = false, int DV = DP>
__global__ void __launch_bounds__(NW * 64, (NW == 8 || DP == 256) ? 1 : 2)
fwd_prefill_kernel(const KParams p, const int n_mblocks) {
    static_assert(!PACK || (NW == 4 && !DROP), "packed grouped heads: 4-wave geometry, no dropout");
    constexpr bool DIRECT = LD == 3;                 // contiguous K/V, loaded straight into LDS
    constexpr bool PDIRECT = LD == 4;                // paged K/V (per-page descriptors), loaded straight into LDS
    constexpr bool TO_LDS = DIRECT || PDIRECT;
    constexpr bool FAST = LD == 1 || DIRECT, PAGED_BUF = LD == 2 || PDIRECT;
    using Cfg = PrefillCfg<DP, BF16, NW>;
    constexpr int NT = Cfg::NT, BM = Cfg::BM, CPR = Cfg::CPR, CH = Cfg::CH;
    // DV <= DP: the head dims this instantiation serves, rounded up to 32.  The tiles keep the
    // layout of DP columns (zero beyond the head dim); the products stop at column DV.
    static_assert(DV % 32 == 0 && DV <= DP && 2 * DV > DP, "DV");
    constexpr int KS = DV / 16;   // k-steps of QK^T
    constexpr int NDB = DV / 32;  // 32-wide d blocks of O
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    FWD_BLK_STAMP(0);

    int bidb, bidh, m_block;
    map_block(p, n_mblocks, 32 * ((NW == 8 || DP == 256) ? 1 : 2), bidb, bidh, m_block);
    const int hk = bidh / p.h_ratio;   // (PACK: p.h_ratio is 1 and bidh is the KV head)
    const int ratio = PACK ? p.pack_gqa : 1;

    const BlockInfo bi(p, bidb);
    const int sq = bi.sq, sk = bi.sk;
    const int sq_rows = sq * ratio;    // rows of this (batch, head): query positions (x grouped heads)
    // Philox state: the caller's values, or (under graph capture) read on the device at replay
    // time from the generator's captured state (at::cuda::philox::unpack, flash_fwd_kernel.h:45)
    unsigned long long ph_seed = p.philox_seed, ph_offset = p.philox_offset;
    if constexpr (DROP) {
        if (p.seed_ptr) {
            ph_seed = (unsigned long long)*p.seed_ptr;
            ph_offset = (unsigned long long)*p.offset_ptr + p.offset_intragraph;
        }
        if (blockIdx.x == 0 && tid == 0 && p.rng_state) {  // flash_fwd_kernel.h:51-54
            p.rng_state[0] = ph_seed;
            p.rng_state[1] = ph_offset;
        }
    }
    if (m_block * BM >= sq_rows) return;
    const unsigned long long drop_off = ph_offset + (unsigned long long)((bidb * p.h + bidh) * 32);  // dropout.h:21

    // ---- key tile range: flash_fwd_kernel.h:59-67 ----
    // query positions [pos_lo_b, pos_hi_b) of the block
    const int pos_lo_b = PACK ? (m_block * BM) / ratio : m_block * BM;
    const int pos_hi_b = PACK ? min(sq, ((m_block + 1) * BM - 1) / ratio + 1) : (m_block + 1) * BM;
    int n_min = !p.is_local ? 0 : max(0, (pos_lo_b + sk - sq - p.wl) / BN);
    int n_max = (sk + BN - 1) / BN;
    if (p.is_causal || p.is_local)
        n_max = min(n_max, (pos_hi_b + sk - sq + p.wr + BN - 1) / BN);

    // Row -> memory.  Unpacked: row = query position, one head.  PACK: (position, head) per row;
    // qg / og / lse_g then point at head 0 of the KV head's group and the head is part of the row offset.
    const int h_all = p.h * ratio;     // query heads of the tensors
    const int head0 = bidh * ratio;
    const unsigned short *qg = (const unsigned short *)p.q + bi.q_off(p.q_bs, p.q_rs, bidb) +
                               (int64_t)head0 * p.q_hs;
    unsigned short *og = (unsigned short *)p.o + bi.q_off(p.o_bs, p.o_rs, bidb) +
                         (int64_t)head0 * p.o_hs;
    float *lse_g = p.lse + ((int64_t)bidb * h_all + head0) * p.seqlen_q;
    float *oacc_g = nullptr;
    if constexpr (SPLIT) {
        const int split = blockIdx.y;
        const int per = (((p.seqlen_k + BN - 1) / BN) + p.num_splits - 1) / p.num_splits;  // from the capacity
        n_min = max(n_min, split * per);
        n_max = min(n_max, (split + 1) * per);
        const int64_t acc_row0 = (((int64_t)split * p.b + bidb) * h_all + head0) * p.seqlen_q;
        lse_g = p.lseaccum + acc_row0;
        oacc_g = p.oaccum + acc_row0 * p.d;
    }
    const int q0w = m_block * BM + wave * 32;  // first row of this wave
    // (position, head of the group) of row `row` of this (batch, head)
    auto row_pos = [&](const int row) __attribute__((always_inline)) -> int { return PACK ? row / ratio : row; };
    auto row_head = [&](const int row) __attribute__((always_inline)) -> int { return PACK ? row % ratio : 0; };
    // element offsets of a row in Q / O, in the LSE plane and in the split partials
    auto q_row_off = [&](const int row) __attribute__((always_inline)) -> int64_t {
        return (int64_t)row_pos(row) * p.q_rs + (PACK ? (int64_t)row_head(row) * p.q_hs : 0);
    };
    auto o_row_off = [&](const int row) __attribute__((always_inline)) -> int64_t {
        return (int64_t)row_pos(row) * p.o_rs + (PACK ? (int64_t)row_head(row) * p.o_hs : 0);
    };
    auto stat_off = [&](const int row) __attribute__((always_inline)) -> int64_t {   // lse_g[...], oacc_g[... * d]
        return PACK ? (int64_t)row_head(row) * p.seqlen_q + row_pos(row) : (int64_t)row;
    };
    const int pos_l = row_pos(q0w + r);                                   // this lane's query position
    const int pos_lo_w = row_pos(q0w), pos_hi_w = PACK ? (q0w + 31) / ratio + 1 : q0w + 32;   // the wave's, [lo, hi)
    FWD_BLK_STAMP(4);

    if (n_max <= n_min) {
        // No visible key for this row block: O = 0, LSE = +inf (flash_fwd_kernel.h:70-106);
        // a split that sees none: partial O = 0, LSE = -inf (:513-552).
        for (int c = tid; c < BM * CPR; c += NT) {
            const int row = m_block * BM + c / CPR, ch = c % CPR;
            if (row < sq_rows && ch * 8 < p.d) {
                if constexpr (SPLIT) {
                    *(f32x4 *)(oacc_g + stat_off(row) * p.d + ch * 8) = f32x4{0, 0, 0, 0};
                    *(f32x4 *)(oacc_g + stat_off(row) * p.d + ch * 8 + 4) = f32x4{0, 0, 0, 0};
                } else {
                    *(i32x4 *)(og + o_row_off(row) + ch * 8) = i32x4{0, 0, 0, 0};
                }
            }
        }
        for (int c = tid; c < BM; c += NT)
            if (m_block * BM + c < sq_rows) lse_g[stat_off(m_block * BM + c)] = SPLIT ? -INFINITY : INFINITY;
        return;
    }

    // ---- K/V addressing ----
    const int bidb_cache = p.cache_batch_idx ? p.cache_batch_idx[bidb] : bidb;
    const unsigned short *kg = (const unsigned short *)p.k + (int64_t)hk * p.k_hs;
    const unsigned short *vg = (const unsigned short *)p.v + (int64_t)hk * p.v_hs;
    const int *bt = nullptr;
    if (p.block_table) {
        bt = p.block_table + (int64_t)bidb * p.bt_bs;
    } else {
        kg += bi.k_off(p.k_bs, p.k_rs, bidb_cache);
        vg += bi.k_off(p.v_bs, p.v_rs, bidb_cache);
    }

    i32x4 kreg[CH], vreg[CH];
    // FAST path state: descriptors and the per-lane byte offset of chunk 0 (out of range for the
    // head-dim padding columns, so that they load zeros)
    const unsigned k_bytes = FAST ? (unsigned)(((int64_t)(sk - 1) * p.k_rs + p.d) * 2) : 0u;
    const unsigned v_bytes = FAST ? (unsigned)(((int64_t)(sk - 1) * p.v_rs + p.d) * 2) : 0u;
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)kg, 0, k_bytes, 0x00020000);
    const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)vg, 0, v_bytes, 0x00020000);
    const bool colok0 = (tid % CPR) * 8 < p.d;
    const unsigned k_voff = colok0 ? (unsigned)(((tid / CPR) * (int)p.k_rs + (tid % CPR) * 8) * 2) : 0x80000000u;
    const unsigned v_voff = colok0 ? (unsigned)(((tid / CPR) * (int)p.v_rs + (tid % CPR) * 8) * 2) : 0x80000000u;
    const unsigned k_rs2 = (unsigned)p.k_rs * 2, v_rs2 = (unsigned)p.v_rs * 2;
    // DIRECT: lane -> (row, chunk) of its LDS slot; the swizzle of a row is unchanged by the row
    // step between a wave's instructions (a multiple of 16 rows at these head dims)
    constexpr int RPI = 64 / CPR;                    // rows per wave instruction
    constexpr int NI = TO_LDS ? BN / (RPI * NW) : 1;  // instructions per wave, tile and tensor
    // (rows RPI * NW apart: a multiple of 16 keeps the lane's swizzle; 8 apart -- d = 256 with 4
    //  waves -- it alternates between two values, for the even and the odd instructions)
    constexpr int NVAR = (RPI * NW) % 16 == 0 ? 1 : 2;
    static_assert(!TO_LDS || (BN % (RPI * NW) == 0 && (RPI * NW) % 8 == 0), "direct LDS loads");
    unsigned dk_voff = 0, dv_voff = 0, dk_voff1 = 0, dv_voff1 = 0;
    // PDIRECT: a wave takes WR consecutive rows of the tile (instruction i the rows RPI * i ...), in
    // groups of DG rows that share a page and so a descriptor.  The swizzle is linear in the row's
    // bits and the wave, instruction and lane parts of the row do not overlap, so a lane's chunk in
    // instruction i is its chunk in instruction 0 XOR a constant.  Chunks in the head-dim padding
    // are redirected to the last real chunk: they meet zero columns of Q (K) or columns of O that
    // are never stored (V), and finite data stays finite.
    constexpr int WR = BN / NW, DG = WR < 16 ? WR : 16, NG = WR / DG, IPG = DG / RPI;
    static_assert(!PDIRECT || (DG % RPI == 0 && NG <= CH), "paged direct LDS loads");
    unsigned pcol0 = 0, prow_k = 0, prow_v = 0;
    const unsigned pcol_max = (unsigned)(p.d / 8 - 1) * 16u;
    if constexpr (PDIRECT) {
        pcol0 = (unsigned)(((lane % CPR) ^ swz<DP>(WR * wave + lane / CPR)) << 4);
        prow_k = (unsigned)(lane / CPR) * k_rs2;
        prow_v = (unsigned)(lane / CPR) * v_rs2;
    }
    if constexpr (DIRECT) {
        const int row0 = RPI * wave + lane / CPR;
        const int ch = (lane % CPR) ^ swz<DP>(row0);
        const bool ok = ch * 8 < p.d;
        dk_voff = ok ? (unsigned)((row0 * (int)p.k_rs + ch * 8) * 2) : 0x80000000u;
        dv_voff = ok ? (unsigned)((row0 * (int)p.v_rs + ch * 8) * 2) : 0x80000000u;
        if constexpr (NVAR == 2) {
            const int ch1 = (lane % CPR) ^ swz<DP>(row0 + RPI * NW);
            const bool ok1 = ch1 * 8 < p.d;
            dk_voff1 = ok1 ? (unsigned)((row0 * (int)p.k_rs + ch1 * 8) * 2) : 0x80000000u;
            dv_voff1 = ok1 ? (unsigned)((row0 * (int)p.v_rs + ch1 * 8) * 2) : 0x80000000u;
        }
    }
    auto load_tile_direct = [&](int n, auto BUFC) __attribute__((always_inline)) {
        constexpr int BUF = decltype(BUFC)::value;
        if constexpr (DIRECT) {
            typedef __attribute__((address_space(3))) void *lds_ptr;
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const unsigned rows = (unsigned)(n * BN + RPI * NW * i);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    k_rsrc, (lds_ptr)(smem + BUF * Cfg::TILE_BYTES + (wave + NW * i) * 1024), 16,
                    ((NVAR == 2 && (i & 1)) ? dk_voff1 : dk_voff) + rows * k_rs2, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    v_rsrc, (lds_ptr)(smem + (2 + BUF) * Cfg::TILE_BYTES + (wave + NW * i) * 1024), 16,
                    ((NVAR == 2 && (i & 1)) ? dv_voff1 : dv_voff) + rows * v_rs2, 0, 0, 0);
            }
        }
    };
    // Paged KV: the page ids of a tile are fetched one tile ahead of its rows, so that the
    // block-table lookup is not a second memory latency in front of the K/V loads.
    int pg[FAST ? 1 : CH];
    constexpr int RSTEP = NT / CPR;  // rows between a lane's chunks = rows of one load round
    auto load_pages = [&](int n) {
        if constexpr (PDIRECT) {
            if (n < 0) return;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const int rb = n * BN + WR * wave + DG * g;  // first row of the wave's group g
                pg[g] = bt[(rb < sk ? rb : 0) >> p.page_shift];
            }
        } else if constexpr (PAGED_BUF) {
            if (n < 0) return;
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const int rb = n * BN + RSTEP * i;  // first row of round i: uniform
                // (kept in a vector register until the tile is loaded, one tile later: moving it to
                // a scalar register here would wait for the K/V loads issued just before)
                pg[i] = bt[(rb < sk ? rb : 0) >> p.page_shift];
            }
        } else if constexpr (!FAST) {
            if (!bt || n < 0) return;
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const int grow = n * BN + (tid + i * NT) / CPR;
                const int safe = grow < sk ? grow : 0;
                pg[i] = bt[p.page_shift >= 0 ? safe >> p.page_shift : safe / p.page_size];
            }
        }
    };
    auto load_tile_pdirect = [&](int n, auto BUFC) __attribute__((always_inline)) {
        constexpr int BUF = decltype(BUFC)::value;
        if constexpr (PDIRECT) {
            typedef __attribute__((address_space(3))) void *lds_ptr;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const int rb = n * BN + WR * wave + DG * g;
                const int inpage = rb & (p.page_size - 1);
                const int pgi = __builtin_amdgcn_readfirstlane(pg[g]);
                const int valid = min(sk - rb, DG);  // rows of this group that exist
                const unsigned kb_ = valid > 0 ? (unsigned)(((int64_t)(valid - 1) * p.k_rs + p.d) * 2) : 0u;
                const unsigned vb_ = valid > 0 ? (unsigned)(((int64_t)(valid - 1) * p.v_rs + p.d) * 2) : 0u;
                const auto kr_ = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(kg + (int64_t)pgi * p.k_bs + (int64_t)inpage * p.k_rs), 0, kb_, 0x00020000);
                const auto vr_ = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(vg + (int64_t)pgi * p.v_bs + (int64_t)inpage * p.v_rs), 0, vb_, 0x00020000);
#pragma unroll
                for (int j = 0; j < IPG; j++) {
                    const int i = g * IPG + j;
                    const unsigned col = min(pcol0 ^ (unsigned)(swz<DP>(RPI * i) << 4), pcol_max);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        kr_, (lds_ptr)(smem + BUF * Cfg::TILE_BYTES + (wave * NI + i) * 1024), 16,
                        col + prow_k + (unsigned)(RPI * j) * k_rs2, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        vr_, (lds_ptr)(smem + (2 + BUF) * Cfg::TILE_BYTES + (wave * NI + i) * 1024), 16,
                        col + prow_v + (unsigned)(RPI * j) * v_rs2, 0, 0, 0);
                }
            }
        }
    };
    auto load_tile = [&](int n) {
        if constexpr (FAST) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                kreg[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    k_rsrc, k_voff + (unsigned)(n * BN + RSTEP * i) * k_rs2, 0, 0));
                vreg[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    v_rsrc, v_voff + (unsigned)(n * BN + RSTEP * i) * v_rs2, 0, 0));
            }
            return;
        }
        if constexpr (PAGED_BUF) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const int rb = n * BN + RSTEP * i;
                const int inpage = rb & (p.page_size - 1);
                const int pgi = __builtin_amdgcn_readfirstlane(pg[i]);
                const int valid = min(sk - rb, p.page_size - inpage);  // rows of this round that exist
                const unsigned kb_ = valid > 0 ? (unsigned)(((int64_t)(valid - 1) * p.k_rs + p.d) * 2) : 0u;
                const unsigned vb_ = valid > 0 ? (unsigned)(((int64_t)(valid - 1) * p.v_rs + p.d) * 2) : 0u;
                const auto kr_ = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(kg + (int64_t)pgi * p.k_bs + (int64_t)inpage * p.k_rs), 0, kb_, 0x00020000);
                const auto vr_ = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(vg + (int64_t)pgi * p.v_bs + (int64_t)inpage * p.v_rs), 0, vb_, 0x00020000);
                kreg[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(kr_, k_voff, 0, 0));
                vreg[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(vr_, v_voff, 0, 0));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int c = tid + i * NT;
            const int row = c / CPR, ch = c % CPR;
            const int grow = n * BN + row;
            const bool ok = grow < sk && ch * 8 < p.d;
            int64_t ko, vo;
            if (bt) {  // paged: csrc/flash_attn/src/utils.h:297-315; page ids fetched a tile ahead
                const int safe = ok ? grow : 0;
                const int inpage = p.page_shift >= 0 ? safe & (p.page_size - 1) : safe % p.page_size;
                ko = (int64_t)pg[i] * p.k_bs + (int64_t)inpage * p.k_rs;
                vo = (int64_t)pg[i] * p.v_bs + (int64_t)inpage * p.v_rs;
            } else {
                ko = (int64_t)grow * p.k_rs;
                vo = (int64_t)grow * p.v_rs;
            }
            kreg[i] = ok ? *(const i32x4 *)(kg + ko + ch * 8) : i32x4{0, 0, 0, 0};
            vreg[i] = ok ? *(const i32x4 *)(vg + vo + ch * 8) : i32x4{0, 0, 0, 0};
        }
    };
    // LDS byte offsets are loop invariant per lane: computing the XOR swizzle inside the tile loop
    // costs more VALU issue slots than the softmax itself, so they are built once here and the
    // tile loop is unrolled by two, which makes the buffer / k-step / key-block parts compile-time
    // constants that fold into the DS instructions' immediate offset field.
    int st_off[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) {
        const int c = tid + i * NT;
        st_off[i] = lds_off<DP>(c / CPR, c % CPR);
    }
    auto store_tile = [&](auto BUFC) {
        constexpr int BUF = decltype(BUFC)::value;
#pragma unroll
        for (int i = 0; i < CH; i++) {
            *(i32x4 *)(smem + st_off[i] + BUF * Cfg::TILE_BYTES) = kreg[i];
            *(i32x4 *)(smem + st_off[i] + (2 + BUF) * Cfg::TILE_BYTES) = vreg[i];
        }
    };

    // 8 waves (one workgroup per CU, nothing overlaps its prologue): the first tile is requested
    // before anything else and the Q rows share its memory round trip (measured with the workgroup
    // timeline: prologue 5.45 -> 4.5 us per workgroup at b8 s4096 h32; with two 4-wave workgroups
    // per CU the old order is level or ahead).
    constexpr bool QIMG = TO_LDS && NW == 8;
    auto request_first_tile = [&]() __attribute__((always_inline)) {
        if constexpr (DIRECT) {
            load_tile_direct(n_max - 1, std::integral_constant<int, 0>{});
        } else if constexpr (PDIRECT) {
            load_pages(n_max - 1);
            load_tile_pdirect(n_max - 1, std::integral_constant<int, 0>{});
            if (n_max - 2 >= n_min) load_pages(n_max - 2);
        }
    };
    if constexpr (QIMG) request_first_tile();

    // ---- Q fragments (B operand of S^T = K Q^T): lane (r, hh) holds Q[q0w + r][16 s + 8 hh ..] ----
    // Read in that layout from global memory, a request covers 32 bytes of each of 32 rows, which
    // keeps the address unit busy for microseconds per workgroup; with the direct-to-LDS tile loads
    // the wave's rows instead go as coalesced 1-KiB requests into a private LDS image (lane ->
    // (row, chunk) of its slot; the swizzle is linear in the row's bits, so the chunk of request i
    // is that of request 0 XOR a constant) and the fragments are read from there.
    i32x4 qf[KS];
    const bool q_image = QIMG && (int64_t)(p.seqlen_q + 256) * p.q_rs * 2 < ((int64_t)1 << 30);
    if (q_image) {
        typedef __attribute__((address_space(3))) void *lds_ptr;
        const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)qg, 0, (unsigned)(((int64_t)(sq - 1) * p.q_rs + p.d) * 2), 0x00020000);
        char *fb = smem + (wave < 2 ? Cfg::TILE_BYTES + wave * Cfg::QIMG_BYTES
                           : wave < 4 ? 3 * Cfg::TILE_BYTES + (wave - 2) * Cfg::QIMG_BYTES
                                      : Cfg::KV_LDS + (wave - 4) * Cfg::QIMG_BYTES);
        const int c0_ = (lane % CPR) ^ swz<DP>(lane / CPR);
        const unsigned lrow = (unsigned)(lane / CPR) * (unsigned)p.q_rs * 2;
#pragma unroll
        for (int i = 0; i < 32 / RPI; i++) {
            const int ch = c0_ ^ swz<DP>(RPI * i);
            const unsigned voff = ch * 8 < p.d ? (unsigned)ch * 16 + lrow + (unsigned)(q0w + RPI * i) * (unsigned)p.q_rs * 2
                                               : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rsrc, (lds_ptr)(fb + i * 1024), 16, voff, 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the wave's Q rows and its part of the first tile are in LDS
#pragma unroll
        for (int s = 0; s < KS; s++) qf[s] = *(const i32x4 *)(fb + lds_off<DP>(r, 2 * s + hh));
        // (read out before the barrier that precedes the first request into the second buffers)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
        const int row = q0w + r;
        const bool rok = row < sq_rows;
        const unsigned short *qr = qg + (rok ? q_row_off(row) : 0);
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int col = 16 * s + 8 * hh;
            qf[s] = (rok && col < p.d) ? *(const i32x4 *)(qr + col) : i32x4{0, 0, 0, 0};
        }
    }

    f32x16 o[NDB];
#pragma unroll
    for (int i = 0; i < NDB; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) o[i][j] = 0.f;
    float m_run = -INFINITY;  // reference point of the exponentials (raw score units): the running
                              // row max, followed lazily; same in both lane halves
    float l_run = 0.f;        // running row sum, PARTIAL per lane half

    float alibi = 0.f;
    const bool has_alibi = p.alibi_slopes != nullptr;
    // (PACK: the slope of the lane's own head)
    if (has_alibi) alibi = p.alibi_slopes[(int64_t)bidb * p.alibi_bs + head0 + row_head(q0w + r)] / p.scale;
    // ALiBi in LINEAR form.  Under a causal mask, or a window that ends at the diagonal or before,
    // every visible key has col <= row + sk - sq, so the bias -slope |row + sk - sq - col| is
    // slope * col plus a constant of the row, and a constant of the row cancels in the softmax.
    // The kernel then adds alibi * (col - cref), cref = the block's first diagonal column (small
    // magnitudes), which is one multiply-add against a compile-time key offset per score, and
    // puts the row constant back into the returned LSE: + slope * cref in causal mode, where the
    // reference itself adds slope * col (mask.h:180-181), - slope * (row + sk - sq - cref) under
    // a window (mask.h:183).
    const bool alibi_lin = has_alibi && (p.is_causal || (p.is_local && p.wr <= 0));
    const int cref = pos_lo_b + sk - sq;
    const float c2 = p.scale_log2;

    // per-lane LDS read offsets (see the note at st_off)
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
    int k_rd[KS];        // K row read of k-step ks: row r (+32 per key block), chunk 2 ks + hh
    int v_rd[NDB][2];    // V transposed read of d block db: rows 4 hh + tr_q (+8), +16 per k-step
#pragma unroll
    for (int ks = 0; ks < KS; ks++) k_rd[ks] = lds_off<DP>(r, 2 * ks + hh);
#pragma unroll
    for (int db = 0; db < NDB; db++)
#pragma unroll
        for (int e = 0; e < 2; e++)
            v_rd[db][e] = lds_off<DP>(4 * hh + tr_q + 8 * e, 4 * db + 2 * tr_g + (tr_p >> 1)) + 8 * (tr_p & 1);
    constexpr int ROWB = DP * 2;

    if constexpr (TO_LDS) {
        if constexpr (!QIMG) request_first_tile();
        FWD_BLK_STAMP(5);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's part of the first tile is in LDS
        FWD_BLK_STAMP(6);
    } else {
        load_pages(n_max - 1);
        load_tile(n_max - 1);
        if (n_max - 2 >= n_min) load_pages(n_max - 2);
        store_tile(std::integral_constant<int, 0>{});
    }
    __syncthreads();
    FWD_BLK_STAMP(1);

#ifdef VFA_STAMP
    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
    const unsigned long long t_begin_ = last_;
#endif
    // one K/V tile; BUF (compile time) is the LDS buffer it sits in
    auto tile_step = [&](const int n, auto BUFC) {
        constexpr int BUF = decltype(BUFC)::value;
        const bool has_next = n > n_min;
        STAMP(7);
        if (has_next) {
            if constexpr (DIRECT) {
                load_tile_direct(n - 1, std::integral_constant<int, BUF ^ 1>{});
            } else if constexpr (PDIRECT) {
                load_tile_pdirect(n - 1, std::integral_constant<int, BUF ^ 1>{});
                if (n - 2 >= n_min) load_pages(n - 2);
            } else {
                load_tile(n - 1);
                if (n - 2 >= n_min) load_pages(n - 2);
            }
        }
        STAMP(0);

        const char *kb = smem + BUF * Cfg::TILE_BYTES;
        const char *vb = smem + (2 + BUF) * Cfg::TILE_BYTES;
        const int c0 = n * BN;
        // The tile range [n_min, n_max) is the block's; a wave whose 32 rows see none of this
        // tile's keys (above its own diagonal, or left of its window) only helps with the staging.
        bool skip = false;
        if (p.is_causal || p.is_local) skip = c0 >= pos_hi_w + sk - sq + p.wr;
        if (p.is_local) skip = skip || (c0 + BN <= pos_lo_w + sk - sq - p.wl);
        if constexpr (PACK) skip = skip || q0w >= sq_rows;   // a wave past the last row (short sequences of a ragged batch)
        if (!skip) {

        // ---- S^T = K Q^T : s[kb2][reg] = score(key = n*64 + 32 kb2 + (reg&3) + 8 (reg>>2) + 4 hh,
        //                                      query = q0w + r) ----
        // Vector instructions of a wave run in the shadow of its OWN MFMAs (not of those of the
        // other wave on the SIMD), so softmax work is placed next to MFMAs it does not depend on:
        // the maximum of the first 32 keys and the exponentials of the first 16, taken against the
        // current reference point, go with the MFMAs of the second 32 keys.  They stand unless the
        // tile needs a mask or moves the reference point (both rare), which redoes them.
        bool need_mask = c0 + BN > sk;
        if (p.is_causal || p.is_local) need_mask = need_mask || (c0 + BN > pos_lo_w + 1 + sk - sq + p.wr);
        if (p.is_local) need_mask = need_mask || (c0 < pos_hi_w - 1 + sk - sq - p.wl);
        const bool need = need_mask || has_alibi;
        f32x16 s[2];
        i32x4 pf[4];
        float ps[4] = {0.f, 0.f, 0.f, 0.f};  // four short row-sum chains instead of one of 32 adds
        auto exp_step = [&](const int ks, const float mc_) __attribute__((always_inline)) {
            const int b8 = 8 * (ks & 1);   // k-step ks = 16 keys = regs 8(ks&1).. of s[ks>>1]
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[ks >> 1][b8 + 2 * w], c2, -mc_));
                const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[ks >> 1][b8 + 2 * w + 1], c2, -mc_));
                ps[w] += e0 + e1;
                pf[ks][w] = pack2<BF16>(e0, e1);
            }
        };
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const i32x4 kf = *(const i32x4 *)(kb + k_rd[ks]);
            s[0] = MMA_S(kf, ks, s[0]);
        }
        constexpr bool SPEC = DP >= 128 && !DROP;  // measured: +1..2 % at d = 128, -2.5 % at d = 64
        const float mc_old = (m_run == -INFINITY ? 0.f : m_run) * c2;
        float mx = -INFINITY;
        if (!SPEC || need) {  // (a masked tile takes its exponentials after the mask)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const i32x4 kf = *(const i32x4 *)(kb + k_rd[ks] + 32 * ROWB);
                s[1] = MMA_S(kf, ks, s[1]);
            }
        } else {
            // second 32 keys: MFMA ks, then a slice of the vector work on the first 32 keys, fenced
            // so that the scheduler keeps each slice behind its MFMA; K fragments two steps ahead
            i32x4 kf2[3];
            kf2[0] = *(const i32x4 *)(kb + k_rd[0] + 32 * ROWB);
            kf2[1] = *(const i32x4 *)(kb + k_rd[1] + 32 * ROWB);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                if (ks + 2 < KS) kf2[(ks + 2) % 3] = *(const i32x4 *)(kb + k_rd[ks + 2] + 32 * ROWB);
                s[1] = MMA_S(kf2[ks % 3], ks, s[1]);
                if constexpr (!SPEC) continue;
#pragma unroll
                for (int j = ks * 16 / KS; j < (ks + 1) * 16 / KS; j++) mx = fmaxf(mx, s[0][j]);
#pragma unroll
                for (int pr = 0; pr < 8; pr++) {  // pair pr: k-step pr>>2, dword pr&3 of its P fragment
                    if (pr * KS / 8 != ks) continue;
                    const int st = pr >> 2, w = pr & 3, b8 = 8 * (st & 1);
                    const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[0][b8 + 2 * w], c2, -mc_old));
                    const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[0][b8 + 2 * w + 1], c2, -mc_old));
                    ps[w] += e0 + e1;
                    pf[st][w] = pack2<BF16>(e0, e1);
                    // (keeps the slice here: LLVM would sink it to the block of its first use)
                    asm volatile("" : "+v"(pf[st][w]), "+v"(ps[w]));
                }
                asm volatile("" : "+v"(mx));
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        STAMP(1);
        // ---- masks and ALiBi, only where needed (mask.h:128-209) ----
        // Each form sits behind a branch of its own that the usual tile walks around.  As
        // alternatives of one if / else chain they made the compiler give the join its own 32
        // registers and move every score into them on the path that needs no mask at all
        // (16 v_mov_b64 per tile: 5-6 % of the forward's time at d = 64 and d = 128).
        if (alibi_lin && !need_mask) {
            // interior tile: bias only
            const float t = alibi * (float)(c0 + 4 * hh - cref);
#pragma unroll
            for (int kb2 = 0; kb2 < 2; kb2++)
#pragma unroll
                for (int j = 0; j < 16; j++)
                    s[kb2][j] = __builtin_fmaf(alibi, (float)(32 * kb2 + (j & 3) + 8 * (j >> 2)), s[kb2][j] + t);
        }
        if (need_mask && !has_alibi && !p.is_local) {
            // causal diagonal and / or key tail only: key offset of register j (a constant) against
            // one per-lane limit -- a compare and a select per score
            const int hi = p.is_causal ? min(sk, pos_l + 1 + sk - sq) : sk;
            const int lim = hi - c0 - 4 * hh;
#pragma unroll
            for (int kb2 = 0; kb2 < 2; kb2++)
#pragma unroll
                for (int j = 0; j < 16; j++)
                    s[kb2][j] = (32 * kb2 + (j & 3) + 8 * (j >> 2) < lim) ? s[kb2][j] : -INFINITY;
        }
        if ((need_mask && (has_alibi || p.is_local)) || (has_alibi && !alibi_lin && !need_mask)) {
            const int row = pos_l;
            const int lo = p.is_local ? max(0, row + sk - sq - p.wl) : 0;
            const int hi = (p.is_causal || p.is_local) ? min(sk, row + 1 + sk - sq + p.wr) : sk;
#pragma unroll
            for (int kb2 = 0; kb2 < 2; kb2++)
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int col = c0 + 32 * kb2 + (j & 3) + 8 * (j >> 2) + 4 * hh;
                    float x = s[kb2][j];
                    if (has_alibi)
                        x = alibi_lin ? x + alibi * (float)(col - cref)
                                      : x - alibi * fabsf((float)(row + sk - sq - col));
                    s[kb2][j] = (col >= lo && col < hi) ? x : -INFINITY;
                }
        }
        if constexpr (SPEC) {
            if (need) {   // the early maximum was taken on scores without mask or bias
                mx = s[0][0];
#pragma unroll
                for (int j = 1; j < 16; j++) mx = fmaxf(mx, s[0][j]);
            }
        }
        if constexpr (!SPEC) {
#pragma unroll
            for (int j = 0; j < 16; j++) mx = fmaxf(mx, s[0][j]);
        }

        // ---- online softmax (softmax.h:135-166), one query row per lane ----
#pragma unroll
        for (int j = 0; j < 16; j++) mx = fmaxf(mx, s[1][j]);
        mx = half_swap_max(mx);
        // The reference point of the exponentials follows the running maximum lazily: it moves
        // (for every row of the wave at once) only when some row's maximum outgrew it by more
        // than 2^RESCALE_LOG2, so P stays below that bound and the rescale of O becomes rare.
        // Any reference point gives the same O and LSE up to rounding.
        constexpr float RESCALE_LOG2 = DROP ? 0.f : 4.f;
        const bool upd = __any((mx - m_run) * c2 > RESCALE_LOG2);
        const float m_new = upd ? fmaxf(m_run, mx) : m_run;
        const float m_cur = (m_new == -INFINITY) ? 0.f : m_new;
        if (upd) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_cur) * c2);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < NDB; i++)
#pragma unroll
                for (int j = 0; j < 16; j++) o[i][j] *= alpha;
        }
        m_run = m_new;
        const float mc = m_cur * c2;
        if constexpr (SPEC) {
            if (need || upd) {  // the early exponentials were taken on unmasked scores or an old reference
#pragma unroll
                for (int w = 0; w < 4; w++) ps[w] = 0.f;
                exp_step(0, mc);
                exp_step(1, mc);
            }
        } else {
            exp_step(0, mc);
        }

        if constexpr (DROP) {
            exp_step(1, mc); exp_step(2, mc); exp_step(3, mc);
            unsigned short *sd = nullptr;
            if (p.s_dmask)
                sd = (unsigned short *)p.s_dmask +
                     (((int64_t)bidb * p.h + bidh) * p.sq_rounded_p + q0w + r) * p.sk_rounded_p + c0 + 4 * hh;
#pragma unroll
            for (int kb2 = 0; kb2 < 2; kb2++) {
                const unsigned keep = keep_bits_qlane(ph_seed, drop_off, q0w >> 4, (c0 >> 5) + kb2,
                                                      r, hh, p.drop_keep_u8);
#pragma unroll
                for (int e = 0; e < 2; e++)
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const int ks = 2 * kb2 + e, j = 8 * e + 2 * w;
                        const unsigned k0 = (keep >> j) & 1u, k1 = (keep >> (j + 1)) & 1u;
                        // sign bit = dropped (dropout.h:31-33); registers j, j+1 are adjacent keys
                        if (sd)
                            *(unsigned *)(sd + 32 * kb2 + (j & 3) + 8 * (j >> 2)) =
                                (unsigned)pf[ks][w] ^ ((k0 ^ 1u) << 15) ^ ((k1 ^ 1u) << 31);
                        pf[ks][w] &= (int)((k0 ? 0xffffu : 0u) | (k1 ? 0xffff0000u : 0u));
                    }
            }
        }
        STAMP(2);
        // ---- O^T += V^T P^T.  P fragment of k-step ks (16 keys) = regs 8(ks&1).. of s[ks>>1];
        // element j of lane half hh is key 16 ks + 8 (j>>2) + 4 hh + (j&3). ----
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            if constexpr (SPEC) { if (ks < 2) exp_step(ks + 2, mc); }  // next to the MFMAs of k-step ks
            else if constexpr (!DROP) { if (ks < 3) exp_step(ks + 1, mc); }
#pragma unroll
            for (int db = 0; db < NDB; db++) {
                const s16x4 v0 = lds_read_tr16(vb + v_rd[db][0] + 16 * ks * ROWB);
                const s16x4 v1 = lds_read_tr16(vb + v_rd[db][1] + 16 * ks * ROWB);
                i32x4 vf;
                const i32x2 a0 = __builtin_bit_cast(i32x2, v0), a1 = __builtin_bit_cast(i32x2, v1);
                vf[0] = a0[0]; vf[1] = a0[1]; vf[2] = a1[0]; vf[3] = a1[1];
                o[db] = MMA_O(vf, ks, o[db]);
            }
        }
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        STAMP(3);
        }  // !skip

        if constexpr (TO_LDS) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) before the barrier
        else if (has_next) store_tile(std::integral_constant<int, BUF ^ 1>{});
        STAMP(4);
        __syncthreads();
        STAMP(5);
    };

    for (int n = n_max - 1;;) {
        tile_step(n, std::integral_constant<int, 0>{});
        if (--n < n_min) break;
        tile_step(n, std::integral_constant<int, 1>{});
        if (--n < n_min) break;
    }
    FWD_BLK_STAMP(2);

#ifdef VFA_STAMP
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long *dst = g_stamps + ((size_t)blockIdx.x * NW + wave) * 8;
        for (int i = 0; i < 6; i++) dst[i] = acc_[i];
        dst[6] = last_ - t_begin_;
        dst[7] = (unsigned long long)(n_max - n_min);
    }
#endif
    // ---- epilogue: normalise, LSE (softmax.h:168-185), store O through LDS as whole rows ----
    const float l_tot = half_swap_sum(l_run);
    const bool bad = (l_tot == 0.f) || (l_tot != l_tot);
    const float inv = (bad ? 1.f : 1.f / l_tot) * (DROP ? p.rp_dropout : 1.f);  // softmax.h:181
    float lse = bad ? (SPLIT ? -INFINITY : INFINITY) : m_run * p.scale + __logf(l_tot);
    if (alibi_lin && !bad) {  // the row constant left out of the linear ALiBi form
        const float slope = alibi * p.scale;
        lse += p.is_causal ? slope * (float)cref : -slope * (float)(pos_l + sk - sq - cref);
    }
    if (hh == 0 && q0w + r < sq_rows) lse_g[stat_off(q0w + r)] = lse;

    if constexpr (SPLIT) {
        // fp32 partial result straight from the accumulators: lane (r, hh) holds, per d block and
        // group g4, four consecutive columns of row q0w + r
        if (q0w + r < sq_rows) {
            float *dst = oacc_g + stat_off(q0w + r) * p.d;
#pragma unroll
            for (int db = 0; db < NDB; db++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int col = 32 * db + 8 * g4 + 4 * hh;
                    if (col < p.d)
                        *(f32x4 *)(dst + col) = f32x4{o[db][4 * g4] * inv, o[db][4 * g4 + 1] * inv,
                                                      o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv};
                }
        }
        return;
    }
    // (the loop's last __syncthreads() already fenced all LDS reads of the K/V tiles)
    char *ob = smem + wave * 32 * Cfg::O_ROWB;
#pragma unroll
    for (int db = 0; db < NDB; db++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            i32x2 w;
            w[0] = pack2<BF16>(o[db][4 * g4 + 0] * inv, o[db][4 * g4 + 1] * inv);
            w[1] = pack2<BF16>(o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv);
            *(i32x2 *)(ob + r * Cfg::O_ROWB + (32 * db + 8 * g4 + 4 * hh) * 2) = w;
        }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's own LDS writes
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < (32 * CPR) / 64; i++) {
        const int c = lane + 64 * i;
        const int row = c / CPR, ch = c % CPR;
        const i32x4 w = *(const i32x4 *)(ob + row * Cfg::O_ROWB + ch * 16);
        if (q0w + row < sq_rows && ch * 8 < p.d)
            *(i32x4 *)(og + o_row_off(q0w + row) + ch * 8) = w;
    }
    FWD_BLK_STAMP(3);
}

template <int DP, bool BF16, int NW, int LD, bool DROP = false, bool SPLIT = false, bool PACK = false, int DV = DP>
static hipError_t launch_prefill_impl(const KParams &p, hipStream_t stream) {
    using Cfg = PrefillCfg<DP, BF16, NW>;
    const int n_mblocks = (p.seqlen_q * (PACK ? p.pack_gqa : 1) + Cfg::BM - 1) / Cfg::BM;
    const long long blocks = (long long)n_mblocks * p.b * p.h;   // (PACK: p.h is the KV head count)
    if (blocks <= 0) return hipSuccess;
    auto kern = fwd_prefill_kernel<DP, BF16, NW, LD, DROP, SPLIT, PACK, DV>;
    static std::atomic<uint64_t> lds_set{0};
    if (hipError_t e = set_dyn_lds_once((const void *)kern, Cfg::LDS_BYTES, lds_set); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, SPLIT ? (unsigned)p.num_splits : 1u), dim3(Cfg::NT),
                       Cfg::LDS_BYTES, stream, p, n_mblocks);
    return hipGetLastError();
}

// Contiguous K/V whose per-(batch, head) extent fits a 32-bit buffer offset take the buffer-load path.
static bool kv_fits_buffer(const KParams &p) {
    if (p.block_table) return false;
    const int64_t lim = (int64_t)1 << 30;  // bytes, with room for the tile offsets added per lane
    const int64_t rows = (int64_t)p.seqlen_k + 256;
    return rows * p.k_rs * 2 < lim && rows * p.v_rs * 2 < lim;
}

// Paged K/V with power-of-two pages that hold whole load rounds (NT / CPR rows) take per-page
// buffer descriptors.
template <int DP, int NW>
static bool paged_fits_buffer(const KParams &p) {
    constexpr int RSTEP = NW * 64 / (DP / 8);
    const int64_t lim = (int64_t)1 << 30;
    return p.block_table && p.page_shift >= 0 && p.page_size % RSTEP == 0 &&
           (int64_t)p.page_size * p.k_rs * 2 < lim && (int64_t)p.page_size * p.v_rs * 2 < lim;
}

// ... straight into LDS: one descriptor per wave and group of at most 16 rows.
template <int DP>
static bool paged_fits_direct(const KParams &p) {
    const int64_t lim = (int64_t)1 << 30;
    return p.block_table && p.page_shift >= 0 && p.page_size % 16 == 0 &&
           (int64_t)p.page_size * p.k_rs * 2 < lim && (int64_t)p.page_size * p.v_rs * 2 < lim;
}

#if (!defined(VFA_DROPOUT_TU) && !defined(VFA_SPLIT_TU)) || defined(VFA_FWD64_TU) || defined(VFA_FWD96_TU)
template <int DP, bool BF16, int NW, int DV = DP>
static hipError_t launch_prefill(const KParams &p, hipStream_t stream) {
    // contiguous K/V: straight into LDS where the lane map allows it (measured on MI355X against
    // register staging: +2..4 % at seqlen 4096, +9 % at 512..2048, d = 64 +4 %)
    if (kv_fits_buffer(p)) return launch_prefill_impl<DP, BF16, NW, 3, false, false, false, DV>(p, stream);
    // paged K/V through per-page descriptors: straight into LDS (measured against register staging,
    // b4 s4096 h16 causal: page 256 d = 64 633 -> 742, d = 128 854 -> 879 TFLOP/s; above d = 128 the
    // staging registers do not fit).  Development switch 5 = 1 takes register staging.
    if (paged_fits_direct<DP>(p) && !(g_tuning[5] == 1 && DP <= 128 && paged_fits_buffer<DP, NW>(p)))
        return launch_prefill_impl<DP, BF16, NW, 4, false, false, false, DV>(p, stream);
#ifdef VFA_DEV   // register staging of paged tiles: reachable only through development switch 5
    if constexpr (DP <= 128)
        if (paged_fits_buffer<DP, NW>(p)) return launch_prefill_impl<DP, BF16, NW, 2, false, false, false, DV>(p, stream);
#endif
    return launch_prefill_impl<DP, BF16, NW, 0, false, false, false, DV>(p, stream);
}

#endif
#if defined(VFA_PACK_TU)
// Grouped query heads packed into the rows of their KV head (template parameter PACK), a translation
// unit of its own (fwd_prefill_pack.hip).  Every head dim, contiguous K/V or pages of a power of two;
// with or without a split of the keys.
bool fwd_prefill_pack_ok(const KParams &p) {
    if (p.h_ratio <= 1) return false;
    return kv_fits_buffer(p) || paged_fits_direct<128>(p);   // (the page test does not depend on the head dim)
}
template <int DP, bool BF16, bool SPLIT>
static hipError_t launch_pack(const KParams &p, hipStream_t stream) {
    return kv_fits_buffer(p) ? launch_prefill_impl<DP, BF16, 4, 3, false, SPLIT, true>(p, stream)
                             : launch_prefill_impl<DP, BF16, 4, 4, false, SPLIT, true>(p, stream);
}
template <int DP, bool BF16>
static hipError_t launch_pack_s(const KParams &p, hipStream_t stream) {
    return p.num_splits > 1 ? launch_pack<DP, BF16, true>(p, stream) : launch_pack<DP, BF16, false>(p, stream);
}
hipError_t run_fwd_prefill_pack(const KParams &p_in, hipStream_t stream) {
    KParams p = p_in;
    p.pack_gqa = p_in.h_ratio;   // the grid walks the KV heads
    p.h = p_in.h_k;
    p.h_ratio = 1;
    const bool bf = p.is_bf16;
    if (p.d <= 64) return bf ? launch_pack_s<64, true>(p, stream) : launch_pack_s<64, false>(p, stream);
    if (p.d <= 128) return bf ? launch_pack_s<128, true>(p, stream) : launch_pack_s<128, false>(p, stream);
    return bf ? launch_pack_s<256, true>(p, stream) : launch_pack_s<256, false>(p, stream);
}
#elif defined(VFA_FWD64_TU)
// The d <= 64 instantiations, a translation unit of their own (fwd_prefill_d64.hip) because they are
// built with -fno-slp-vectorize: hipcc packs pairs of the softmax's fp32 adds and multiplies into
// v_pk_add_f32 / v_pk_mul_f32, which hold the vector issue port longer than the two instructions
// they replace -- and at this head dim the vector instructions, not the MFMAs, bound the tile
// (measured b4 s2048 h16 causal 589 -> 605 TFLOP/s, b8 s2048 h32 full 774 -> 808, b4 s4096 h32
// causal 776 -> 806; the d = 128 kernels lose 5-11 % under the same flag).
hipError_t run_fwd_prefill_64(const KParams &p, hipStream_t stream) {
    return p.is_bf16 ? launch_prefill<64, true, 4>(p, stream) : launch_prefill<64, false, 4>(p, stream);
}
#elif defined(VFA_FWD96_TU)
// Head dims 65..96, a translation unit of their own (fwd_prefill_d96.hip): the d = 128 tile layout
// with the products cut off at column 96 (6 of 8 k-steps of Q K^T, 3 of 4 column blocks of P V; the
// skipped ones multiply zero padding, so the results are the same bits).  Measured b8 s4096 h32 d96
// causal 1.120 -> 1.042 ms, full 1.915 -> 1.831, d80 causal 1.034 -> 0.978.
hipError_t run_fwd_prefill_96(const KParams &p, hipStream_t stream, bool waves8) {
    const bool bf = p.is_bf16;
    if (waves8) return bf ? launch_prefill<128, true, 8, 96>(p, stream) : launch_prefill<128, false, 8, 96>(p, stream);
    return bf ? launch_prefill<128, true, 4, 96>(p, stream) : launch_prefill<128, false, 4, 96>(p, stream);
}
#elif !defined(VFA_DROPOUT_TU) && !defined(VFA_SPLIT_TU)
// Tile choice per head dim (the counterpart of flash_fwd_launch_template.h:168-370).
hipError_t run_fwd_prefill_64(const KParams &p, hipStream_t stream);          // fwd_prefill_d64.hip
hipError_t run_fwd_prefill_96(const KParams &p, hipStream_t stream, bool waves8);   // fwd_prefill_d96.hip
bool fwd_prefill_w64_ok(const KParams &p);
hipError_t run_fwd_prefill_w64(const KParams &p, hipStream_t stream);

hipError_t run_fwd_prefill_wide(const KParams &p, hipStream_t stream);       // fwd_prefill_wide.hip
long long fwd_prefill_persist_items(const KParams &p, int waves_override);   // fwd_prefill_persist.hip
hipError_t run_fwd_prefill_persist(const KParams &p, hipStream_t stream, int waves_override);

hipError_t run_fwd_prefill(const KParams &p, hipStream_t stream) {
    const bool bf = p.is_bf16;
#ifdef VFA_DEV
    if (g_tuning[4] == 1 && fwd_prefill_w64_ok(p)) return run_fwd_prefill_w64(p, stream);   // 64 rows per wave: 3-8 % behind
    // persistent workgroups at d <= 64: level with or behind the 4-wave block kernel (two 4-wave
    // workgroups per CU already overlap each other's prologues: b4 s2048 h16 causal 0.063 vs 0.064 ms,
    // b8 s2048 h32 full 0.382 vs 0.357; one 8-wave workgroup per CU 5-15 % behind)
    if (p.d <= 64 && g_tuning[3] == 2 && fwd_prefill_persist_ok(p))
        return run_fwd_prefill_persist(p, stream, g_tuning[6] == 1 ? 8 : 0);
#endif
    // the geometry: vfa_dispatch.h (work per CU first, keys per block second)
    const int geo = fwd_geometry(p, device_num_cus());
    if (p.d <= 64) return run_fwd_prefill_64(p, stream);
    if (geo == VFA_GEO_WIDE32) return run_fwd_prefill_wide(p, stream);
    if (p.d > 128) return bf ? launch_prefill<256, true, 4>(p, stream) : launch_prefill<256, false, 4>(p, stream);
    // Persistent workgroups (fwd_prefill_persist.hip) wherever every CU gets at least two 256-row
    // blocks to chain: the seams between a CU's blocks then cost no memory round trip.  Measured on
    // MI355X against the block kernels (interleaved, bf16, d = 128): b8 s4096 h32 causal 892 -> 961
    // TFLOP/s, full 1013 -> 1077; b4 s8192 h32/8 causal 1010 -> 1060; b32 s512 h16 causal 477 -> 529;
    // b16 s1024 h16 full 869 -> 960.
    if (geo == VFA_GEO_PERSIST8) return run_fwd_prefill_persist(p, stream, 0);
    if (p.d <= 96) return run_fwd_prefill_96(p, stream, geo == VFA_GEO_BLOCK8);
    if (geo == VFA_GEO_BLOCK8) return bf ? launch_prefill<128, true, 8>(p, stream) : launch_prefill<128, false, 8>(p, stream);
    return bf ? launch_prefill<128, true, 4>(p, stream) : launch_prefill<128, false, 4>(p, stream);
}

#elif defined(VFA_SPLIT_TU)
// Split-KV variants (4 waves; contiguous or buffer-paged K/V), a translation unit of their own.
template <int DP, bool BF16, int DV = DP>
static hipError_t launch_prefill_split(const KParams &p, hipStream_t stream) {
    return kv_fits_buffer(p) ? launch_prefill_impl<DP, BF16, 4, 3, false, true, false, DV>(p, stream)
                             : launch_prefill_impl<DP, BF16, 4, 4, false, true, false, DV>(p, stream);
}
// Whether the split variants can load this K/V layout (else the caller takes the decode kernel).
bool fwd_prefill_split_ok(const KParams &p) {
    if (kv_fits_buffer(p)) return true;
    return p.d <= 64 ? paged_fits_direct<64>(p) : (p.d <= 128 ? paged_fits_direct<128>(p) : paged_fits_direct<256>(p));
}
hipError_t run_fwd_prefill_split(const KParams &p, hipStream_t stream) {
    const bool bf = p.is_bf16;
    if (p.d <= 64) return bf ? launch_prefill_split<64, true>(p, stream) : launch_prefill_split<64, false>(p, stream);
    if (p.d <= 96) return bf ? launch_prefill_split<128, true, 96>(p, stream) : launch_prefill_split<128, false, 96>(p, stream);   // (see run_fwd_prefill_96)
    if (p.d <= 128) return bf ? launch_prefill_split<128, true>(p, stream) : launch_prefill_split<128, false>(p, stream);
    return bf ? launch_prefill_split<256, true>(p, stream) : launch_prefill_split<256, false>(p, stream);
}
#else   // VFA_DROPOUT_TU: the dropout variants are compiled as a translation unit of their own
// Dropout is a training feature: one geometry (4 waves) per head dim; contiguous K/V that fit a
// 32-bit buffer offset load straight into LDS at d <= 128, anything else by generic loads.
hipError_t run_fwd_prefill_dropout(const KParams &p, hipStream_t stream) {
    const bool bf = p.is_bf16;
    const bool direct = kv_fits_buffer(p);
    if (p.d <= 64) {
        if (direct) return bf ? launch_prefill_impl<64, true, 4, 3, true>(p, stream) : launch_prefill_impl<64, false, 4, 3, true>(p, stream);
        return bf ? launch_prefill_impl<64, true, 4, 0, true>(p, stream)
                  : launch_prefill_impl<64, false, 4, 0, true>(p, stream);
    }
    if (p.d <= 128) {
        if (direct) return bf ? launch_prefill_impl<128, true, 4, 3, true>(p, stream) : launch_prefill_impl<128, false, 4, 3, true>(p, stream);
        return bf ? launch_prefill_impl<128, true, 4, 0, true>(p, stream)
                  : launch_prefill_impl<128, false, 4, 0, true>(p, stream);
    }
    return bf ? launch_prefill_impl<256, true, 4, 0, true>(p, stream)
              : launch_prefill_impl<256, false, 4, 0, true>(p, stream);
}
#endif  // VFA_DROPOUT_TU

}  // namespace vfa

#if defined(VFA_STAMP) && !defined(VFA_DROPOUT_TU)
extern "C" __attribute__((visibility("default"))) int vfa_debug_read_blk_fwd(unsigned long long *host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vfa::g_blk_fwd), n * sizeof(unsigned long long));
}
extern "C" __attribute__((visibility("default"))) int vfa_debug_read_stamps(unsigned long long *host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vfa::g_stamps), n * sizeof(unsigned long long));
}
#endif
