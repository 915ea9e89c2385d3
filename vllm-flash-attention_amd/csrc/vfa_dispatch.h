// Which kernel geometry a problem takes: the counterpart of the reference's per-head-dim tile table
// (csrc/flash_attn/src/flash_fwd_launch_template.h:168-370) and of its "enough CTAs to fill the
// SMs?" test (csrc/flash_attn/flash_api.cpp:158-219).  Host code, shared by the launchers and by
// vfa_dispatch_geometry() (include/vfa.h), so that what a call takes can be asked and tested.
//
// The choices are functions of the WORK PER CU first -- how many 256-row (or 256-key) blocks the
// launch has against the chip's CUs -- and of the keys a block sees second:
//   * two or more blocks per CU (one, for the forward without a mask), on a problem the persistent
//     kernels take: one workgroup per CU walks its blocks (VFA_GEO_PERSIST8);
//   * fewer 256-row blocks than CUs: 128-row blocks of 4 waves, two per CU (VFA_GEO_BLOCK4) -- the
//     launch then has twice the workgroups and fills twice the CUs.  Measured (MI355X, bf16 d = 128,
//     b1 h8 s4096): forward 0.117 -> 0.096 ms, dK/dV 0.189 -> 0.157, dQ (full attention) 0.140 -> 0.123;
//   * between one and two blocks per CU under a mask: 128-row blocks too -- a CU with ONE causal
//     block has nothing to balance its weight against (b1 h8 s8192 causal: forward 0.221 -> 0.193 ms,
//     dK/dV 0.379 -> 0.340, dQ 0.280 -> 0.256);
//   * otherwise the crossovers measured in keys per block: 8 waves (VFA_GEO_BLOCK8) share a K/V tile
//     between twice the rows, 4 waves split a causal diagonal more finely and overlap each other's
//     prologues.
#pragma once
#include <algorithm>
#include <cstdint>

#include "../../include/vfa.h"
#include "vfa_kernel_params.h"

namespace vfa {

bool fwd_prefill_persist_ok(const KParams &p);   // fwd_prefill_persist.hip
bool fwd_prefill_wide_ok(const KParams &p);      // fwd_prefill_wide.hip
bool bwd_dq_persist_ok(const KParams &p);        // bwd_dq_persist.hip
bool bwd_dkdv_w64_ok(const KParams &p);          // bwd.hip
bool bwd_dkdv_wide_ok(const KParams &p);         // bwd.hip

// keys a row block can see under a sliding window: the window's span, not the sequence
inline int64_t visible_span(const KParams &p) {
    int64_t span = p.seqlen_k;
    if (p.is_local)
        span = std::min<int64_t>(span, (int64_t)(p.wl < 0 ? p.seqlen_k : p.wl) + (p.wr < 0 ? p.seqlen_k : p.wr) + 1);
    return span;
}
inline long long row_blocks_256(const KParams &p) { return (long long)p.b * p.h * ((p.seqlen_q + 255) / 256); }
inline long long key_blocks_256(const KParams &p) {
    return (long long)p.b * p.h_k * ((p.seqlen_k + 255) / 256) * std::max(1, p.hsplit);
}

// Forward (fwd_prefill_kernel / fwd_prefill_persist_kernel), no dropout.
inline int fwd_geometry(const KParams &p, const int cus) {
    if (p.d <= 64) return VFA_GEO_BLOCK4;   // softmax-bound, 4 waves everywhere (+8..40 %)
    // d > 128: one wave per SIMD; the hand-pipelined kernel wherever it runs (development switch 4 = 2: never)
    if (p.d > 128) return (g_tuning[4] != 2 && fwd_prefill_wide_ok(p)) ? VFA_GEO_WIDE32 : VFA_GEO_BLOCK4;
    const long long blocks = row_blocks_256(p);
    const bool masked = p.is_causal || p.is_local;
    // (without a mask the walk is ahead of the block kernel from one block per CU: b1 h8 s8192 0.262 -> 0.253 ms)
    // (head dims up to 96, whose kernels stop their products at column 96: from seqlen 4096 the block
    //  kernels are 2-3 % ahead -- b8 s4096 h32 d96 causal 1.044 -> 1.021 ms, full 1.785 -> 1.733, b4
    //  s8192 1.823 -> 1.785; at 2048 the walk is level (full) or 3.5 % ahead (causal), at 1024 9 % ahead)
    const bool walk_pays = !(p.d <= 96 && p.seqlen_q >= 4096);
    if (g_tuning[3] == 2 ? fwd_prefill_persist_ok(p)
                         : (g_tuning[3] != 1 && walk_pays && fwd_prefill_persist_ok(p) && blocks >= (masked ? 2LL : 1LL) * cus))
        return VFA_GEO_PERSIST8;
    if (g_tuning[6]) return g_tuning[6] == 1 ? VFA_GEO_BLOCK8 : VFA_GEO_BLOCK4;   // development switch 6
    if (blocks < cus || (masked && blocks < 2LL * cus)) return VFA_GEO_BLOCK4;
    // measured crossovers (16k..64k tokens): full attention 8 waves from seqlen 512 (+3..9 %);
    // under a mask from 3072 visible keys (4 waves +1 % at 2048, 8 waves +2 % at 3072, +5 % at 16k)
    const bool big = masked ? (p.seqlen_q >= 3072 && visible_span(p) >= 3072) : p.seqlen_q >= 512;
    return big ? VFA_GEO_BLOCK8 : VFA_GEO_BLOCK4;
}

// dQ (bwd_dq_kernel / bwd_dq_persist_kernel), no dropout.
inline int dq_geometry(const KParams &p, const int cus) {
    if (p.d <= 64 || p.d > 128) return VFA_GEO_BLOCK4;   // d = 64: 4 waves win everywhere measured
    const long long blocks = row_blocks_256(p);
    if (g_tuning[3] == 2 ? bwd_dq_persist_ok(p) : (g_tuning[3] != 1 && bwd_dq_persist_ok(p) && blocks >= 2LL * cus))
        return VFA_GEO_PERSIST8;
    if (g_tuning[7]) return g_tuning[7] == 1 ? VFA_GEO_BLOCK8 : VFA_GEO_BLOCK4;   // development switch 7
    const bool masked = p.is_causal || p.is_local;
    if (blocks < cus || (masked && blocks < 2LL * cus)) return VFA_GEO_BLOCK4;
    // under a mask 4 waves win up to seqlen 4096 and 8 waves from 8192; without one 8 waves from 2048;
    // under a window the span of visible keys counts (below 3072: 4 waves)
    const bool big = (masked ? p.seqlen_q >= 6144 : p.seqlen_q >= 2048) && (!p.is_local || visible_span(p) >= 3072);
    return big ? VFA_GEO_BLOCK8 : VFA_GEO_BLOCK4;
}

// dK/dV (bwd_dkdv_kernel / bwd_dkdv_w64_kernel / bwd_dkdv_wide_kernel), no dropout.
inline int dkdv_geometry(const KParams &p, const int cus) {
    // d > 128: one workgroup for all 256 columns (8 products per key block and slice) wherever it
    // runs; else the head dim cut in two column halves (12 products; development switch 0 = 5)
    if (p.d > 128) return (g_tuning[0] != 5 && bwd_dkdv_wide_ok(p)) ? VFA_GEO_WIDE32 : VFA_GEO_BLOCK4;
    const bool w64_ok = bwd_dkdv_w64_ok(p);
    if (g_tuning[0] == 2 && w64_ok) return VFA_GEO_WAVE64;
    if (g_tuning[0] == 1) return VFA_GEO_BLOCK8;
    if (g_tuning[0] == 5) return VFA_GEO_BLOCK4;
    if (p.d <= 64) return p.seqlen_k <= 128 ? VFA_GEO_BLOCK4 : VFA_GEO_BLOCK8;   // 4 waves 15-30 % behind at d = 64
    // d = 128: 128-key blocks of 4 waves (two per CU) while the 256-key blocks cannot fill the chip,
    // and up to 1024 keys (b32 s512 h16 causal 0.153 -> 0.139 ms; level at 2048, 1 % behind from there)
    const long long kblocks = key_blocks_256(p);
    if (kblocks < cus || p.seqlen_k <= 1024) return VFA_GEO_BLOCK4;
    if ((p.is_causal || p.is_local) && kblocks < 2LL * cus) return VFA_GEO_BLOCK4;   // one masked block per CU
    // 64 keys per wave at one wave per SIMD: full attention over long key ranges (+1-5 % from seqlen
    // 4096; at 2048 between -3 and +2 %).  Under a causal mask its waves cannot skip the pieces above
    // the diagonal, four 64-row steps per 256-key block: 7 % behind at seqlen 4096, 2 % at 8192, 3 %
    // ahead at 16384 (b1 h32 3.757 -> 3.648 ms).
    // (head dims up to 96 stay with the 8-wave kernel, which stops its products at column 96)
    if (w64_ok && g_tuning[0] == 0 && !p.is_local && p.d > 96) {
        const int from = p.is_causal ? 12288 : 3072;
        if (p.seqlen_k >= from && p.seqlen_q >= from) return VFA_GEO_WAVE64;
    }
    return VFA_GEO_BLOCK8;
}

}  // namespace vfa
