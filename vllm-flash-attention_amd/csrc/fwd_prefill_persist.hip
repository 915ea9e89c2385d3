// FlashAttention-2 forward for gfx950 / MI355X: PERSISTENT workgroups.
//
// Same tile loop as fwd_prefill_kernel (8 waves x 32 query rows, 64-key tiles straight into LDS,
// S^T = K Q^T with the query on the lane; csrc/flash_attn/src/flash_fwd_kernel.h:27-470), but one
// workgroup per CU stays resident and walks a list of (batch, head, 256-row block) items instead
// of the reference's one-CTA-per-row-block launch (flash_fwd_launch_template.h:51-94).  With one
// 8-wave workgroup per CU nothing overlaps a block's prologue (kernel arguments, block mapping,
// the first K/V tile's and the Q rows' memory round trip) and epilogue: 5-6 us per block against a
// 75 us tile loop at seqlen 4096 and a 12 us loop at 512.  Here the K/V tiles of consecutive blocks
// form ONE stream through the two LDS buffers: during the last tile of a block the workgroup
// requests the first tile of the next block and every wave its next 32 Q rows (a private LDS
// image), so the seam costs the epilogue's vector work and nothing waits for memory.
//
// Work distribution is static (no counters, nothing to reset between launches, graph-replay safe):
// positions of the launch order with L % 8 == x belong to the 32 workgroups with blockIdx % 8 == x
// (one XCD under round-robin dispatch -- a locality assumption, not a correctness one), whole
// (batch, kv head) pairs per XCD, heaviest blocks first within groups of about 64 positions (see
// map_block_at).  Workgroup c of an XCD takes position c of every even round of 32 and position
// 31 - c of every odd round: under a causal mask the cost falls linearly inside a group, so a
// forward and a reversed round add up to the same cost for every workgroup.
//
// Scope: dense batches (no cu_seqlens / block table / cache indirection), d <= 128, causal or full
// attention with key / row tails; everything else stays with fwd_prefill_kernel.
#include "vfa_kernel_params.h"
#include <type_traits>

namespace vfa {

// Geometries: d <= 128 with 8 waves (256-row blocks, one workgroup per CU) and d <= 64 with 4 waves
// (128-row blocks, two workgroups per CU: at d = 64 the softmax outweighs the MFMAs and two
// independent workgroups overlap their phases better than one of twice the size).
template <int DP_, int NW_>
struct PersistCfg {
    static constexpr int DP = DP_, NW = NW_, NT = NW * 64, BM = NW * 32, BN = 64, CPR = DP / 8;
    static constexpr int WG_PER_CU = NW == 4 ? 2 : 1;
    static constexpr int TILE_BYTES = BN * DP * 2;
    static constexpr int KV_LDS = 4 * TILE_BYTES;               // K0 K1 V0 V1
    static constexpr int QIMG_BYTES = 32 * DP * 2;              // a wave's 32 Q rows
    static constexpr int OST_BYTES = 32 * 128;                  // a wave's O rows, 64 columns at a time
    static constexpr int OFF_Q = KV_LDS;
    static constexpr int OFF_O = OFF_Q + NW * QIMG_BYTES;
    static constexpr int LDS_BYTES = OFF_O + NW * OST_BYTES;
    static_assert(LDS_BYTES * WG_PER_CU <= 160 * 1024, "LDS per CU");
};

template <int DP_, bool BF16, int NW_, int DV = DP_>
__global__ void __launch_bounds__(NW_ * 64, (PersistCfg<DP_, NW_>::WG_PER_CU))
fwd_prefill_persist_kernel(const KParams p, const int n_mblocks, const int n_items) {
    using Cfg = PersistCfg<DP_, NW_>;
    constexpr int DP = Cfg::DP, NW = Cfg::NW, NT = Cfg::NT, BM = Cfg::BM, BN = Cfg::BN, CPR = Cfg::CPR;
    // DV <= DP: the head dims this instantiation serves, rounded up to 32 (tiles keep the layout
    // of DP columns, the products stop at column DV: the rest would multiply zero padding)
    static_assert(DV % 32 == 0 && DV <= DP && 2 * DV > DP, "DV");
    constexpr int KS = DV / 16, NDB = DV / 32;
    constexpr int RPI = 64 / CPR;                 // rows per wave instruction of a direct-to-LDS load
    constexpr int NI = BN / (RPI * NW);           // instructions per wave, tile and tensor
    constexpr int ROWB = DP * 2;
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The later-dispatched wave of each SIMD loses the issue arbitration (by age) in every phase;
    // a static priority for that half, set once, measured 0.4-1.1 % less time.
    if (NW_ > 4 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int r = lane & 31, hh = lane >> 5;
    const int sq = p.seqlen_q, sk = p.seqlen_k;   // dense batch: every entry has these lengths
    const float c2 = p.scale_log2;

    // ---- this workgroup's walk through the launch order ----
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwx = gridDim.x >> 3;
    const int n_x = (n_items - xcd + 7) >> 3;     // positions L = 8 s + xcd < n_items
    const int rounds = (n_x + nwx - 1) / nwx;

    struct Blk { int bidb, bidh, m_block, n_max; };
    // Advances `t` to this workgroup's next block with visible keys; blocks that see none (causal,
    // more query rows than keys) get O = 0, LSE = +inf here (flash_fwd_kernel.h:70-106).
    auto next_item = [&](int &t, Blk &b) __attribute__((always_inline)) -> bool {
        for (;;) {
            if (++t >= rounds) return false;
            const int s = t * nwx + ((t & 1) ? nwx - 1 - wg : wg);
            if (s >= n_x) continue;
            map_block_at(p, n_mblocks, 32 * Cfg::WG_PER_CU, 8 * s + xcd, b.bidb, b.bidh, b.m_block);
            b.n_max = (sk + BN - 1) / BN;
            if (p.is_causal) b.n_max = min(b.n_max, ((b.m_block + 1) * BM + sk - sq + BN - 1) / BN);
            if (b.n_max > 0) return true;
            unsigned short *og = (unsigned short *)p.o + (int64_t)b.bidb * p.o_bs + (int64_t)b.bidh * p.o_hs;
            float *lse_g = p.lse + ((int64_t)b.bidb * p.h + b.bidh) * sq;
            for (int c = tid; c < BM * CPR; c += NT) {
                const int row = b.m_block * BM + c / CPR, ch = c % CPR;
                if (row < sq && ch * 8 < p.d) *(i32x4 *)(og + (int64_t)row * p.o_rs + ch * 8) = i32x4{0, 0, 0, 0};
            }
            for (int c = tid; c < BM; c += NT)
                if (b.m_block * BM + c < sq) lse_g[b.m_block * BM + c] = INFINITY;
        }
    };

    // ---- loop-invariant per-lane offsets ----
    // direct-to-LDS tile loads: lane -> (row, chunk) of its slot in a wave instruction's 1 KiB; the
    // lane fetches the chunk that the XOR swizzle places there (rows of a wave's two instructions
    // are 32 apart: same swizzle)
    const unsigned k_rs2 = (unsigned)p.k_rs * 2, v_rs2 = (unsigned)p.v_rs * 2, q_rs2 = (unsigned)p.q_rs * 2;
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
    // a wave's 32 Q rows as eight coalesced 1-KiB requests into its private LDS image (the swizzle
    // is linear in the row's bits: the chunk of request i is that of request 0 XOR a constant)
    char *const qimg = smem + Cfg::OFF_Q + wave * Cfg::QIMG_BYTES;
    const int qc0 = (lane % CPR) ^ swz<DP>(lane / CPR);
    const unsigned q_lrow = (unsigned)(lane / CPR) * q_rs2;
    auto load_qimg = [&](const Blk &b) __attribute__((always_inline)) {
        const unsigned short *qg = (const unsigned short *)p.q + (int64_t)b.bidb * p.q_bs + (int64_t)b.bidh * p.q_hs;
        const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)qg, 0, q_bytes, 0x00020000);
        const unsigned row0 = (unsigned)(b.m_block * BM + wave * 32);
        // (opaque: the eight per-lane offsets below are loop invariants the compiler would otherwise
        //  keep in registers through the tile loop)
        int qc = qc0;
        unsigned ql = q_lrow;
        asm volatile("" : "+v"(qc), "+v"(ql));
#pragma unroll
        for (int i = 0; i < 32 / RPI; i++) {
            const int ch = qc ^ swz<DP>(RPI * i);
            const unsigned voff = ch * 8 < p.d ? (unsigned)ch * 16 + ql + (row0 + RPI * i) * q_rs2 : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rsrc, (lds_ptr)(qimg + i * 1024), 16, voff, 0, 0, 0);
        }
    };
    // LDS read offsets (built once: the swizzle costs more vector issue slots than the softmax)
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
    int k_rd[KS], v_rd[NDB][2];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) k_rd[ks] = lds_off<DP>(r, 2 * ks + hh);
#pragma unroll
    for (int db = 0; db < NDB; db++)
#pragma unroll
        for (int e = 0; e < 2; e++)
            v_rd[db][e] = lds_off<DP>(4 * hh + tr_q + 8 * e, 4 * db + 2 * tr_g + (tr_p >> 1)) + 8 * (tr_p & 1);

    // ---- first block ----
    Blk cur, nxt;
    int t = -1;
    if (!next_item(t, cur)) return;
    auto k_rsrc = kv_rsrc(cur, p.k, p.k_bs, p.k_hs, k_bytes);
    auto v_rsrc = kv_rsrc(cur, p.v, p.v_bs, p.v_hs, v_bytes);
    load_tile(k_rsrc, v_rsrc, cur.n_max - 1, std::integral_constant<int, 0>{});
    load_qimg(cur);
    bool have_next = next_item(t, nxt);   // (decoded under the flight of the loads)

    i32x4 qf[KS];
    f32x16 o[NDB];
    float m_run, l_run;
    int n, q0w;
    auto begin_block = [&]() __attribute__((always_inline)) {
        // the wave's Q rows have landed (vmcnt(0) precedes every call): B-operand fragments of
        // S^T = K Q^T, lane (r, hh) holds Q[q0w + r][16 s + 8 hh ..]
#pragma unroll
        for (int s = 0; s < KS; s++) qf[s] = *(const i32x4 *)(qimg + lds_off<DP>(r, 2 * s + hh));
#pragma unroll
        for (int i = 0; i < NDB; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) o[i][j] = 0.f;
        m_run = -INFINITY;   // reference point of the exponentials: the running row max, followed lazily
        l_run = 0.f;         // running row sum, partial per lane half
        n = cur.n_max - 1;
        q0w = cur.m_block * BM + wave * 32;
        // (the image is requested anew during this block's last tile, which may be its first)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    begin_block();
    __syncthreads();

    // ---- normalise, LSE (softmax.h:168-185), O through LDS as whole 128-byte half rows ----
    char *const ost = smem + Cfg::OFF_O + wave * Cfg::OST_BYTES;
    auto epilogue = [&]() __attribute__((always_inline)) {
        const float l_tot = half_swap_sum(l_run);
        const bool bad = (l_tot == 0.f) || (l_tot != l_tot);
        const float inv = bad ? 1.f : 1.f / l_tot;
        const float lse = bad ? INFINITY : m_run * p.scale + __logf(l_tot);
        float *lse_g = p.lse + ((int64_t)cur.bidb * p.h + cur.bidh) * sq;
        if (hh == 0 && q0w + r < sq) lse_g[q0w + r] = lse;
        unsigned short *og = (unsigned short *)p.o + (int64_t)cur.bidb * p.o_bs + (int64_t)cur.bidh * p.o_hs;
        // staging rows of 128 bytes, 16-byte chunks XOR-swizzled with bits 1..3 of the row: the
        // 8-byte writes (lane = row) and the 16-byte reads (8 lanes per row) both spread over all banks
        const int wsw = (r >> 1) & 7;
#pragma unroll
        for (int half = 0; half < DP / 64; half++) {
#pragma unroll
            for (int d2 = 0; d2 < 2; d2++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int db = 2 * half + d2;
                    if (db >= NDB) continue;   // (a column block beyond DV: never stored)
                    i32x2 w;
                    w[0] = pack2<BF16>(o[db][4 * g4 + 0] * inv, o[db][4 * g4 + 1] * inv);
                    w[1] = pack2<BF16>(o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv);
                    *(i32x2 *)(ost + r * 128 + (((4 * d2 + g4) ^ wsw) << 4) + 8 * hh) = w;
                }
            // (the wave's own staging rows; a wave's LDS operations execute in order)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = (lane >> 3) + 8 * i, ch = lane & 7;
                const i32x4 w = *(const i32x4 *)(ost + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
                const int col = 64 * half + 8 * ch;
                if (q0w + row < sq && col < p.d) *(i32x4 *)(og + (int64_t)(q0w + row) * p.o_rs + col) = w;
            }
        }
    };

    // One K/V tile of the stream; BUF (compile time) is the LDS buffer it sits in.  Returns true
    // after the workgroup's last block.
    auto tile_step = [&](auto BUFC) __attribute__((always_inline)) -> bool {
        constexpr int BUF = decltype(BUFC)::value;
        const bool last = n == 0;
        if (!last) {
            load_tile(k_rsrc, v_rsrc, n - 1, std::integral_constant<int, BUF ^ 1>{});
        } else if (have_next) {
            // the seam: the next block's first tile and Q rows travel under this block's last tile
            k_rsrc = kv_rsrc(nxt, p.k, p.k_bs, p.k_hs, k_bytes);
            v_rsrc = kv_rsrc(nxt, p.v, p.v_bs, p.v_hs, v_bytes);
            load_tile(k_rsrc, v_rsrc, nxt.n_max - 1, std::integral_constant<int, BUF ^ 1>{});
            load_qimg(nxt);
        }

        const char *kb = smem + BUF * Cfg::TILE_BYTES;
        const char *vb = smem + (2 + BUF) * Cfg::TILE_BYTES;
        const int c0 = n * BN;
        // a wave whose 32 rows see none of this tile's keys (above its own diagonal) only helps
        // with the staging
        const bool skip = p.is_causal && c0 >= q0w + 32 + sk - sq;
        if (!skip) {
        // ---- S^T = K Q^T : s[kb2][reg] = score(key = n*64 + 32 kb2 + (reg&3) + 8 (reg>>2) + 4 hh,
        //                                      query = q0w + r) ----
        // The maximum of the first 32 keys and the exponentials of the first 16, taken against the
        // current reference point, go beside the MFMAs of the second 32 keys; a tile that needs a
        // mask or moves the reference point (both rare) redoes them.
        bool need = c0 + BN > sk;
        if (p.is_causal) need = need || (c0 + BN > q0w + 1 + sk - sq);
        f32x16 s[2];
        i32x4 pf[4];
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
        auto exp_step = [&](const int ks, const float mc_) __attribute__((always_inline)) {
            const int b8 = 8 * (ks & 1);
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
            s[0] = mma32<BF16>(kf, qf[ks], ks == 0 ? zero : s[0]);
        }
        constexpr bool SPEC = DP >= 128;   // (measured on the block kernel: +1..2 % at d = 128, -2.5 % at d = 64)
        const float mc_old = (m_run == -INFINITY ? 0.f : m_run) * c2;
        float mx = -INFINITY;
        if (!SPEC || need) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const i32x4 kf = *(const i32x4 *)(kb + k_rd[ks] + 32 * ROWB);
                s[1] = mma32<BF16>(kf, qf[ks], ks == 0 ? zero : s[1]);
            }
        } else {
            i32x4 kf2[3];
            kf2[0] = *(const i32x4 *)(kb + k_rd[0] + 32 * ROWB);
            kf2[1] = *(const i32x4 *)(kb + k_rd[1] + 32 * ROWB);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                if (ks + 2 < KS) kf2[(ks + 2) % 3] = *(const i32x4 *)(kb + k_rd[ks + 2] + 32 * ROWB);
                s[1] = mma32<BF16>(kf2[ks % 3], qf[ks], ks == 0 ? zero : s[1]);
#pragma unroll
                for (int j = ks * 16 / KS; j < (ks + 1) * 16 / KS; j++) mx = fmaxf(mx, s[0][j]);
#pragma unroll
                for (int pr = ks * 8 / KS; pr < (ks + 1) * 8 / KS; pr++) {   // pair pr: k-step pr >> 2, dword pr & 3 of its P fragment
                    const int st = pr >> 2, w = pr & 3, b8 = 8 * (st & 1);
                    const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[0][b8 + 2 * w], c2, -mc_old));
                    const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[0][b8 + 2 * w + 1], c2, -mc_old));
                    ps[w] += e0 + e1;
                    pf[st][w] = pack2<BF16>(e0, e1);
                    asm volatile("" : "+v"(pf[st][w]), "+v"(ps[w]));   // (keeps the slice here)
                }
                asm volatile("" : "+v"(mx));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (need) {
            // causal diagonal and / or key tail: key offset of register j (a constant) against one
            // per-lane limit (mask.h:128-209)
            const int hi = p.is_causal ? min(sk, q0w + r + 1 + sk - sq) : sk;
            const int lim = hi - c0 - 4 * hh;
#pragma unroll
            for (int kb2 = 0; kb2 < 2; kb2++)
#pragma unroll
                for (int j = 0; j < 16; j++)
                    s[kb2][j] = (32 * kb2 + (j & 3) + 8 * (j >> 2) < lim) ? s[kb2][j] : -INFINITY;
            if constexpr (SPEC) {
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
        // the reference point moves (for every row of the wave at once) only when some row's
        // maximum outgrew it by more than 2^4: P stays below that bound, the rescale of O is rare
        const bool upd = __any((mx - m_run) * c2 > 4.f);
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
            if (need || upd) {   // the early exponentials were taken on unmasked scores or an old reference
#pragma unroll
                for (int w = 0; w < 4; w++) ps[w] = 0.f;
                exp_step(0, mc);
                exp_step(1, mc);
            }
        } else {
            exp_step(0, mc);
        }
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            if constexpr (SPEC) { if (ks < 2) exp_step(ks + 2, mc); }   // beside the MFMAs of k-step ks
            else { if (ks < 3) exp_step(ks + 1, mc); }
#pragma unroll
            for (int db = 0; db < NDB; db++) {
                const s16x4 v0 = lds_read_tr16(vb + v_rd[db][0] + 16 * ks * ROWB);
                const s16x4 v1 = lds_read_tr16(vb + v_rd[db][1] + 16 * ks * ROWB);
                const i32x2 a0 = __builtin_bit_cast(i32x2, v0), a1 = __builtin_bit_cast(i32x2, v1);
                const i32x4 vf = {a0[0], a0[1], a1[0], a1[1]};
                o[db] = mma32<BF16>(vf, pf[ks], o[db]);
            }
        }
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        }  // !skip

        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's part of the next tile (and Q image) is in LDS
        __syncthreads();
        if (!last) {
            --n;
            return false;
        }
        epilogue();
        if (!have_next) return true;
        cur = nxt;
        begin_block();
        have_next = next_item(t, nxt);
        return false;
    };

    for (;;) {
        if (tile_step(std::integral_constant<int, 0>{})) break;
        if (tile_step(std::integral_constant<int, 1>{})) break;
    }
}

// Whether a persistent geometry can run this problem (the caller decides whether it should).
bool fwd_prefill_persist_ok(const KParams &p) {
    if (p.d > 128) return false;
    if (p.cu_seqlens_q || p.cu_seqlens_k || p.seqused_k || p.block_table || p.cache_batch_idx || p.knew) return false;
    if (p.alibi_slopes || p.is_local) return false;
    if (p.seqlen_q <= 0 || p.seqlen_k <= 0) return false;
    const int64_t lim = (int64_t)1 << 30;   // per-(batch, head) extents as 32-bit buffer offsets
    const int64_t rq = (int64_t)p.seqlen_q + 256, rk = (int64_t)p.seqlen_k + 256;
    return rq * p.q_rs * 2 < lim && rk * p.k_rs * 2 < lim && rk * p.v_rs * 2 < lim;
}

// Rows per block of the geometry taken for this head dim (waves = rows / 32).
int fwd_prefill_persist_block_rows(const KParams &p, int waves_override) {
    const int nw = waves_override ? waves_override : (p.d <= 64 ? 4 : 8);
    return 32 * nw;
}

// Positions of the launch order: (batch, head, row block).
long long fwd_prefill_persist_items(const KParams &p, int waves_override) {
    const int bm = fwd_prefill_persist_block_rows(p, waves_override);
    return (long long)((p.seqlen_q + bm - 1) / bm) * p.b * p.h;
}

template <int DP, bool BF16, int NW, int DV = DP>
static hipError_t launch_persist(const KParams &p, hipStream_t stream) {
    using Cfg = PersistCfg<DP, NW>;
    const int n_mblocks = (p.seqlen_q + Cfg::BM - 1) / Cfg::BM;
    const long long items = (long long)n_mblocks * p.b * p.h;
    if (items <= 0) return hipSuccess;
    if (items > 0x7fffffffLL) return hipErrorInvalidValue;
    // as many workgroups as stay resident, a multiple of 8 (whole rounds over the XCDs)
    const int grid = std::max(8, device_num_cus() / 8 * 8) * Cfg::WG_PER_CU;
    auto kern = fwd_prefill_persist_kernel<DP, BF16, NW, DV>;
    static std::atomic<uint64_t> lds_set{0};
    if (hipError_t e = set_dyn_lds_once((const void *)kern, Cfg::LDS_BYTES, lds_set); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p, n_mblocks, (int)items);
    return hipGetLastError();
}

hipError_t run_fwd_prefill_persist(const KParams &p, hipStream_t stream, int waves_override) {
    const bool bf = p.is_bf16;
#ifdef VFA_DEV   // measured level with or behind the block kernel: development build only
    if (p.d <= 64) {
        if (waves_override == 8) return bf ? launch_persist<64, true, 8>(p, stream) : launch_persist<64, false, 8>(p, stream);
        return bf ? launch_persist<64, true, 4>(p, stream) : launch_persist<64, false, 4>(p, stream);
    }
#endif
    if (p.d <= 64) return hipErrorInvalidValue;
    // head dims 65..96: 6 of 8 k-steps, 3 of 4 column blocks (b8 s4096 h32 d96 causal 1.120 -> 1.042 ms)
    if (p.d <= 96) return bf ? launch_persist<128, true, 8, 96>(p, stream) : launch_persist<128, false, 8, 96>(p, stream);
    return bf ? launch_persist<128, true, 8>(p, stream) : launch_persist<128, false, 8>(p, stream);
}

}  // namespace vfa
