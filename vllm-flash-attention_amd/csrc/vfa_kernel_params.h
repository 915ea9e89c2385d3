// Device-side parameter block and small helpers shared by the gfx950 attention kernels.
// The fields mirror what the reference passes in Flash_fwd_params / Flash_bwd_params
// (csrc/flash_attn/src/flash.h:24-183); layouts and strides are in ELEMENTS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vfa {

struct KParams {
    const void *q, *k, *v;
    void *o;
    float *lse;                 // (b, h, seqlen_q)
    int64_t q_bs, q_rs, q_hs;   // batch / row / head strides
    int64_t k_bs, k_rs, k_hs;
    int64_t v_bs, v_rs, v_hs;
    int64_t o_bs, o_rs, o_hs;
    int b, h, h_k, h_ratio, d;
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
    int rotary_dim, rotary_interleaved;
    // backward
    const void *dout;
    void *dq, *dk, *dv;
    int64_t do_bs, do_rs, do_hs, dq_bs, dq_rs, dq_hs, dk_bs, dk_rs, dk_hs, dv_bs, dv_rs, dv_hs;
    float *dsoftmax_sum;        // (b, h, seqlen_q_rounded)
    float *dq_accum;            // fp32 (like q, contiguous (b|total, seqlen_q, h, d))
    int seqlen_q_rounded;
    int deterministic;
    int is_bf16;
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

}  // namespace vfa
