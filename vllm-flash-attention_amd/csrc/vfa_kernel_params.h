// Device-side parameter block and small helpers shared by the gfx950 attention kernels.
// The fields mirror what the reference passes in Flash_fwd_params / Flash_bwd_params
// (csrc/flash_attn/src/flash.h:24-183); layouts and strides are in ELEMENTS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace vfa {

// Development switches (kernel-variant selection for A/B timing inside one process).  They exist
// only in the development build of the library (-DVFA_DEV: libvfa_hip_dev.so, which also holds the
// kernels no shipped dispatch takes), where vfa_debug_set_tuning() sets them; in the product build
// the array is a compile-time constant of zeros, every test of it folds away and no call can change
// which kernel a dispatch takes.  All zero = the shipped choices.
//   [0] dK/dV kernel: 0 = auto, 1 = force the 8-wave x 32-key kernel, 2 = force the 4-wave x 64-key one,
//       3 / 4 = two-phase schedule always / never, 5 = 4-wave x 32-key, 6 = persistent workgroups
//   [1] decode kernel: 0 = auto, 1 = always one head per workgroup
//   [2] bit 0 inverts the decode kernel's choice of block-id regrouping over the XCDs, bit 1 turns off
//       the packing of grouped query heads into the rows of their KV head
//   [3] persistent forward / dQ workgroups: 1 = never, 2 = wherever the kernels can run
//   [4] forward at 64 < d <= 128: 1 = the 64-rows-per-wave kernel wherever it applies; at d > 128:
//       2 = never the hand-pipelined kernel (fwd_prefill_wide.hip)
//   [5] paged prefill: 1 = per-page descriptors through staging registers instead of straight into LDS
//   [6] forward at d = 128: 1 = 8 waves, 2 = 4 waves;  [7] the same for the dQ kernel
#ifdef VFA_DEV
extern int g_tuning[8];
#else
constexpr int g_tuning[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

// Compute units of the current device (cached per thread and device).
inline int device_num_cus() {
    static thread_local int cached_dev = -1, cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev != cached_dev) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        cached_dev = dev;
    }
    return cus;
}

// Host side: kernels that need more than 64 KiB of dynamic LDS must be told so once per device.
// `done` is a per-instantiation bit set of the devices already configured (function-local static
// of the launcher), so the runtime call is not repeated on every launch.
inline hipError_t set_dyn_lds_once(const void *kern, int bytes, std::atomic<uint64_t> &done) {
    if (bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (dev < 64 && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev < 64) done.fetch_or(bit, std::memory_order_release);
    return e;
}

struct KParams {
    const void *q, *k, *v;
    void *o;
    float *lse;                 // (b, h, seqlen_q)
    int64_t q_bs, q_rs, q_hs;   // batch / row / head strides
    int64_t k_bs, k_rs, k_hs;
    int64_t v_bs, v_rs, v_hs;
    int64_t o_bs, o_rs, o_hs;
    int b, h, h_k, h_ratio, d;
    int dbg_flags;
    int pack_gqa;               // rows = (query position, query head of the KV group): decode kernel (a flag),
                                // packed prefill kernel (the group size; p.h = p.h_k, p.h_ratio = 1 there)
    int seqlen_q, seqlen_k;     // max query length, max key length / cache capacity
    const int *cu_seqlens_q;    // b+1 or null
    const int *cu_seqlens_k;    // b+1 cumulative, or b lengths when !k_cumulative, or null
    const int *seqused_k;       // b or null
    int k_cumulative;
    const int *block_table;     // paged KV: (b, bt_bs) page ids, or null
    int64_t bt_bs;
    int page_size;
    int page_shift;             // log2(page_size) when it is a power of two, else -1
    const int *cache_batch_idx; // b or null
    const float *alibi_slopes;  // (h) or (b,h) or null
    int alibi_bs;               // 0 for (h)
    float scale, scale_log2;
    int is_causal, is_local, wl, wr;
    // split-KV partial results (fp32): oaccum (splits, b, h, seqlen_q, d), lseaccum (splits, b, h, seqlen_q)
    int num_splits;
    float *oaccum, *lseaccum;
    // KV-cache append (+ rotary)
    const void *knew, *vnew;
    int64_t knew_bs, knew_rs, knew_hs, vnew_bs, vnew_rs, vnew_hs;
    int seqlen_knew;
    const void *rotary_cos, *rotary_sin;
    int rotary_dim, rotary_interleaved, seqlen_ro;
    // backward
    const void *dout;
    void *dq, *dk, *dv;
    int64_t do_bs, do_rs, do_hs, dq_bs, dq_rs, dq_hs, dk_bs, dk_rs, dk_hs, dv_bs, dv_rs, dv_hs;
    float *dsoftmax_sum;        // (b, h, seqlen_q_rounded)
    float *dq_accum;            // fp32 (like q, contiguous (b|total, seqlen_q, h, d))
    int seqlen_q_rounded;
    int deterministic;
    int is_bf16;
    // dropout (dropout.h:12-23, flash_api.cpp:105-113)
    int drop_keep_u8;            // keep an entry iff its random byte <= this
    float rp_dropout;            // 1 / (1 - p_dropout)
    float p_keep;                // 1 - p_dropout
    float scale_rp;              // scale * rp_dropout
    unsigned long long philox_seed, philox_offset;
    const long long *seed_ptr, *offset_ptr;   // forward: when non-null the state is read here on the device (graph capture)
    unsigned offset_intragraph;
    unsigned long long *rng_state;   // fwd: written; bwd: read when non-null
    int rng_read;                // backward: take seed / offset from rng_state
    void *s_dmask;               // (b, h, sq_rounded_p, sk_rounded_p) 16-bit or null
    int sq_rounded_p, sk_rounded_p;
    // backward head split: fp32 partial dK / dV slabs (hsplit, part_rows, h_k, d)
    int hsplit;
    float *dk_part, *dv_part;
    int64_t part_stride, part_rows;
};

// Per-batch lengths and offsets: csrc/flash_attn/src/block_info.h:15-34.
struct BlockInfo {
    int sum_s_q, sum_s_k, sq, sk_cache, sk;
    __device__ __forceinline__ BlockInfo(const KParams &p, int bidb) {
        sum_s_q = p.cu_seqlens_q ? p.cu_seqlens_q[bidb] : -1;
        sum_s_k = (p.cu_seqlens_k && p.k_cumulative) ? p.cu_seqlens_k[bidb] : -1;
        sq = p.cu_seqlens_q ? p.cu_seqlens_q[bidb + 1] - sum_s_q : p.seqlen_q;
        sk_cache = !p.cu_seqlens_k ? p.seqlen_k
                                   : (p.k_cumulative ? p.cu_seqlens_k[bidb + 1] - sum_s_k
                                                     : p.cu_seqlens_k[bidb]);
        sk = p.seqused_k ? p.seqused_k[bidb] : sk_cache + (p.knew ? p.seqlen_knew : 0);
    }
    __device__ __forceinline__ int64_t q_off(int64_t bs, int64_t rs, int bidb) const {
        return sum_s_q < 0 ? bidb * bs : (int64_t)sum_s_q * rs;
    }
    __device__ __forceinline__ int64_t k_off(int64_t bs, int64_t rs, int bidb) const {
        return sum_s_k < 0 ? bidb * bs : (int64_t)sum_s_k * rs;
    }
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 32x32x16 MFMA on 16-bit operands held as 4 dwords.
template <bool BF16>
__device__ __forceinline__ f32x16 mma32(i32x4 a, i32x4 b, f32x16 c) {
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                      __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Two fp32 -> one dword of two 16-bit values (round to nearest even), lo in bits 0..15.
template <bool BF16>
__device__ __forceinline__ int pack2(float lo, float hi) {
    if constexpr (BF16) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        bf2 t = {(__bf16)lo, (__bf16)hi};
        return __builtin_bit_cast(int, t);
    } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 t = {(_Float16)lo, (_Float16)hi};
        return __builtin_bit_cast(int, t);
    }
}

template <bool BF16>
__device__ __forceinline__ float to_f32(unsigned short bits) {
    if constexpr (BF16) return __uint_as_float((unsigned)bits << 16);
    else return (float)__builtin_bit_cast(_Float16, bits);
}

// value of lane^32 combined with own by max
__device__ __forceinline__ float half_swap_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ s16x4 lds_read_tr16(const char *smem_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3))) *)(smem_ptr));
}

// XOR swizzle of the 16-byte chunk index of a [rows][DP x 16-bit] LDS tile.  Chosen so that the
// MFMA A-operand row reads (ds_read_b128: 16-lane groups, distinct rows, same chunk) and the
// transposed reads (ds_read_b64_tr_b16: 4 rows x 64 B per 32-lane half) are bank-conflict free.
template <int DP>
__device__ __forceinline__ int swz(int row) {
    if constexpr (DP == 64) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    else return ((row & 3) << 2) | ((row >> 2) & 3);
}
template <int DP>
__device__ __forceinline__ int lds_off(int row, int ch) {
    return row * (DP * 2) + ((ch ^ swz<DP>(row)) << 4);
}

// ---- dropout: Philox-4x32, 7 rounds (csrc/flash_attn/src/philox.cuh:22-49) ----
struct u32x4s { unsigned x, y, z, w; };
__device__ __forceinline__ u32x4s philox7(unsigned long long seed, unsigned sub_lo, unsigned sub_hi,
                                          unsigned long long offset) {
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    unsigned c0 = (unsigned)offset, c1 = (unsigned)(offset >> 32), c2 = sub_lo, c3 = sub_hi;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;  // (the bump after the last round is dead code)
    }
    return u32x4s{c0, c1, c2, c3};
}

// The reference draws one Philox value per lane t (0..31) per 16 x 32 block of the attention
// matrix: subsequence (row / 16, col / 32), offset + (batch * h + head) * 32 + t, and spends its 16
// bytes on rows t/4 and t/4 + 8, columns 2 (t%4) + {0, 1} of the four 8-column atoms; byte
// 4 * atom + 2 * (row >= 8) + (col & 1) decides entry (row, col): kept iff byte <= keep_u8
// (dropout.h:17-23,39-79; utils.h:215-223).  The two helpers below hand each lane of a 32 x 32
// score sub-tile the 16 keep bits of its accumulator registers (bit j <-> register j) at the cost
// of ONE draw per lane: the 64 lanes compute the sub-tile's 64 draws (2 row blocks x 32 lanes)
// once and exchange them.

// Query on the lane (forward, dQ): lane (r = query row in the wave's 32, hh) holds keys
// (j&3) + 8 (j>>2) + 4 hh of the 32-key block.  Its entries come from draws t = 4 (r&7) + 2 hh and
// t + 1 of row block (q0 + r) / 16, half of each; the other halves belong to lane r ^ 8.
__device__ __forceinline__ unsigned keep_bits_qlane(unsigned long long seed, unsigned long long off_bh,
                                                    int row_block0, int col_block, int r, int hh,
                                                    int keep_u8) {
    const int rh = (r >> 3) & 1;
    const int t = 4 * (r & 7) + 2 * hh + rh;
    const u32x4s own = philox7(seed, (unsigned)(row_block0 + (r >> 4)), (unsigned)col_block, off_bh + t);
    unsigned o[4] = {own.x, own.y, own.z, own.w}, x[4];
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = (unsigned)__shfl_xor((int)o[i], 8);
    unsigned keep = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        // draw t_a (registers with (j>>1)&1 == 0) is `own` on the rh == 0 lane, t_b on the rh == 1 lane
        const bool mine = (((j >> 1) & 1) == rh);
        const unsigned dw = mine ? o[j >> 2] : x[j >> 2];
        const unsigned byte = (dw >> (16 * rh + 8 * (j & 1))) & 0xffu;
        keep |= (byte <= (unsigned)keep_u8 ? 1u : 0u) << j;
    }
    return keep;
}

// Key on the lane (dK/dV): lane (r = key in the wave's 32, hh) holds query rows
// (j&3) + 8 (j>>2) + 4 hh of the 32-row sub-block.  Lane L computes draw t = L & 31 of row block
// row_block0 + (L >> 5); the 8 draws a lane needs are fetched through `xch` (1 KiB of LDS private
// to the wave).
__device__ __forceinline__ unsigned keep_bits_klane(unsigned long long seed, unsigned long long off_bh,
                                                    int row_block0, int col_block, int lane, int keep_u8,
                                                    unsigned *xch) {
    const int r = lane & 31, hh = lane >> 5;
    const u32x4s own = philox7(seed, (unsigned)(row_block0 + hh), (unsigned)col_block, off_bh + r);
    __builtin_amdgcn_wave_barrier();
    *(i32x4 *)(xch + 4 * lane) = i32x4{(int)own.x, (int)own.y, (int)own.z, (int)own.w};
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    unsigned keep = 0;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int br = 0; br < 2; br++) {
            const int src = 4 * (a + 4 * hh) + ((r & 7) >> 1) + 32 * br;
            const unsigned dw = xch[4 * src + (r >> 3)];
#pragma unroll
            for (int rh = 0; rh < 2; rh++) {
                const int j = a + 4 * rh + 8 * br;
                const unsigned byte = (dw >> (16 * rh + 8 * (r & 1))) & 0xffu;
                keep |= (byte <= (unsigned)keep_u8 ? 1u : 0u) << j;
            }
        }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    return keep;
}

// block -> (batch, head, m_block).  Blocks b and b+8 share an XCD (round-robin dispatch), so give
// each XCD whole (batch, kv-head) pairs: their K/V then stay in that XCD's L2 while the pair's
// query heads and m-blocks stream through.  Within an XCD the pairs are taken in groups of G, and
// within a group the m-blocks go heaviest first ACROSS its pairs (the last m-block of every pair,
// then the one before, ...): under a causal mask m-block m costs (m + 1) tiles, and pair-by-pair
// order starts the heavy blocks of the next pair only as the light ones of this pair finish, which
// shows as a tail of up to one heavy block per CU.  G covers about two rounds of the XCD's
// `slots` resident workgroups, so the co-resident blocks still come from few pairs.
// L: position in the launch order (the block id of a one-block-per-workgroup launch; a persistent
// workgroup walks several positions).
__device__ __forceinline__ void map_block_at(const KParams &p, const int n_mblocks, const int slots, const int L,
                                             int &bidb, int &bidh, int &m_block) {
    const int g = p.h_ratio;
    const int W = g * n_mblocks;            // blocks per (batch, kv head) pair
    const int P = p.b * p.h_k;
    const int full = (P / 8) * 8 * W;
    int pair, head, mi;
    if (L < full) {
        const int x = L & 7, s = L >> 3;
        const int ppx = P / 8;
        const int G = max(1, min(ppx, (2 * slots + W - 1) / W));
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
__device__ __forceinline__ void map_block(const KParams &p, const int n_mblocks, const int slots, int &bidb,
                                          int &bidh, int &m_block) {
    map_block_at(p, n_mblocks, slots, (int)blockIdx.x, bidb, bidh, m_block);
}


}  // namespace vfa
