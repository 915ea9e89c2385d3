// PyTorch extension module `vllm_flash_attn_2_cuda` for MI355X: the tensor-level boundary the
// reference exposes through pybind (csrc/flash_attn/flash_api.cpp:939-946), re-implemented as a
// shim over the C ABI of libvfa_hip.so (include/vfa.h).  Same function names, argument order,
// return lists and error strings; `bwd` / `varlen_bwd` follow the call sites in
// vllm_flash_attn/flash_attn_interface.py:135-154 and :183-207 (the fork exports no backward).
#include <torch/python.h>
#include <torch/nn/functional.h>
#include <ATen/hip/HIPGeneratorImpl.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <ATen/Parallel.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <optional>
#include <vector>

#include "vfa.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<at::Tensor>;

void expect_shape(const Tensor &t, const char *name, std::initializer_list<int64_t> want) {
    TORCH_CHECK(t.sizes() == at::IntArrayRef(want.begin(), want.size()), name, " must have shape ",
                at::IntArrayRef(want.begin(), want.size()), ", got ", t.sizes());
}
void expect_gpu(const Tensor &t, const char *name) { TORCH_CHECK(t.is_cuda(), name, " must be on CUDA"); }
void expect_contig(const Tensor &t, const char *name) { TORCH_CHECK(t.is_contiguous(), name, " must be contiguous"); }

int dtype_code(const Tensor &q) {
    auto dt = q.scalar_type();
    TORCH_CHECK(dt == at::kHalf || dt == at::kBFloat16, "FlashAttention only support fp16 and bf16 data type");
    return dt == at::kBFloat16 ? VFA_BF16 : VFA_FP16;
}

void run(int rc) { TORCH_CHECK(rc == 0, vfa_last_error()); }

void *current_stream() { return (void *)at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

int cu_count() { return at::cuda::getCurrentDeviceProperties()->multiProcessorCount; }

Tensor pad_head(const Tensor &t, int64_t d) {
    if (d % 8 == 0) return t;
    namespace F = torch::nn::functional;
    return F::pad(t, F::PadFuncOptions({0, 8 - d % 8}));
}

void set_alibi(vfa_params &P, const OptTensor &slopes, int64_t b, int64_t h) {
    if (!slopes.has_value()) return;
    const Tensor &a = slopes.value();
    TORCH_CHECK(a.scalar_type() == at::kFloat, "ALiBi slopes must have dtype fp32");
    expect_gpu(a, "alibi_slopes");
    TORCH_CHECK(a.stride(-1) == 1, "ALiBi slopes tensor must have contiguous last dimension");
    TORCH_CHECK(a.sizes() == at::IntArrayRef({h}) || a.sizes() == at::IntArrayRef({b, h}),
                "alibi_slopes must have shape (num_heads) or (batch_size, num_heads)");
    P.alibi_slopes = a.data_ptr<float>();
    P.alibi_slopes_batch_stride = a.dim() == 2 ? a.stride(0) : 0;
}

// q/k/v/out pointers + strides (set_params_fprop, flash_api.cpp:53-77)
void set_qkvo(vfa_params &P, const Tensor &q, const Tensor &k, const Tensor &v, const Tensor &out,
              bool batched) {
    P.q = q.data_ptr(); P.k = k.data_ptr(); P.v = v.data_ptr(); P.out = out.data_ptr();
    P.q_row_stride = q.stride(-3); P.q_head_stride = q.stride(-2);
    P.k_row_stride = k.stride(-3); P.k_head_stride = k.stride(-2);
    P.v_row_stride = v.stride(-3); P.v_head_stride = v.stride(-2);
    P.o_row_stride = out.stride(-3); P.o_head_stride = out.stride(-2);
    if (batched) {
        P.q_batch_stride = q.stride(0); P.k_batch_stride = k.stride(0);
        P.v_batch_stride = v.stride(0); P.o_batch_stride = out.stride(0);
    }
}

// Split scratch is held by the caller until the launches are enqueued (the reference lets it go
// back to the caching allocator before the launch, flash_api.cpp:212-215).
struct SplitScratch { Tensor lse, out; };
void plan_splits(vfa_params &P, SplitScratch &S, int num_splits, int64_t b, int64_t h, int64_t d,
                 int64_t sk, int64_t sq, const at::TensorOptions &opts) {
    if (num_splits < 1) {
        // Grouped query heads over few query tokens run packed into the rows of their KV head: the
        // library counts the work items the way its decode kernel is launched.
        num_splits = vfa_decode_num_splits(b, h, P.h_k > 0 ? P.h_k : h, d, sk, sq, cu_count());
    }
    TORCH_CHECK(num_splits <= 128, "num_splits > 128 not supported");
    P.num_splits = num_splits;
    if (num_splits > 1) {
        S.lse = torch::empty({num_splits, b, h, sq}, opts.dtype(at::kFloat));
        S.out = torch::empty({num_splits, b, h, sq, d}, opts.dtype(at::kFloat));
        P.softmax_lseaccum = S.lse.data_ptr<float>();
        P.oaccum = S.out.data_ptr<float>();
    }
}

int64_t round128(int64_t x) { return (x + 127) / 128 * 128; }

// Dropout set-up shared by fwd / varlen_fwd (flash_api.cpp:349-354,379-394): the optional
// Return_softmax tensor and the Philox state.  The generator's offset advances by b * h * 32, the
// number of draws the kernels index per (batch, head) (dropout.h:21).
Tensor plan_dropout(vfa_params &P, double p_dropout, bool return_softmax, int64_t b, int64_t h,
                    int64_t sq, int64_t sk, const at::TensorOptions &opts, Tensor &rng_state,
                    std::optional<at::Generator> &gen_) {
    TORCH_CHECK(p_dropout >= 0.0 && p_dropout < 1.0, "p_dropout must be in [0, 1)");
    Tensor p = torch::empty({0}, opts);  // flash_api.cpp:355-357: an empty tensor, not None
    if (return_softmax) {
        TORCH_CHECK(p_dropout > 0.0, "return_softmax is only supported when p_dropout > 0.0");
        // zero-filled (the reference leaves tiles its kernel never visits uninitialised)
        p = torch::zeros({b, h, round128(sq), round128(sk)}, opts);
        P.s_dmask = p.data_ptr();
        P.seqlen_q_rounded = (int)round128(sq);
        P.seqlen_k_rounded = (int)round128(sk);
    }
    P.p_dropout = (float)p_dropout;
    P.rng_state = (uint64_t *)rng_state.data_ptr<int64_t>();
    if (p_dropout > 0.0) {
        auto gen = at::get_generator_or_default<at::CUDAGeneratorImpl>(
            gen_, at::cuda::detail::getDefaultCUDAGenerator());
        std::lock_guard<std::mutex> lock(gen->mutex_);
        // flash_api.cpp:388-394: under graph capture the state is a pair of device pointers plus an
        // offset inside the graph, read by the kernel at replay time; otherwise plain values
        const at::PhiloxCudaState st = gen->philox_cuda_state((uint64_t)(b * h * 32));
        if (st.captured_) {
            P.philox_seed_ptr = st.seed_.ptr;
            P.philox_offset_ptr = st.offset_.ptr;
            P.philox_offset_intragraph = st.offset_intragraph_;
        } else {
            P.philox_seed = st.seed_.val;
            P.philox_offset = st.offset_.val;
        }
    }
    return p;
}


// ---------------------------------------------------------------------------------------------
// fp32 tensors in HOST memory: a plain C++ evaluation of the same function, so that
// flash_attn_func(q, k, v) on CPU fp32 inputs can be checked against torch's
// scaled_dot_product_attention without a GPU (the reference itself rejects such input,
// flash_api.cpp:263-271).  It is taken ONLY for CPU fp32 tensors -- nothing a GPU call could reach:
// CUDA tensors always run the HIP kernels, 16-bit CPU tensors are rejected as in the reference --
// and it is forward-only.  Same conventions as the kernels: bottom-right aligned causal / window
// mask (mask.h:171-172), ALiBi -slope * |i + sk - sq - j|, GQA head i -> i / (h / h_k), empty rows
// -> 0 with LSE = +inf, LSE = max * scale + ln(sum).
std::vector<Tensor> fwd_cpu_fp32(const Tensor &q_, const Tensor &k_, const Tensor &v_, OptTensor &out_,
                                 OptTensor &alibi_slopes_, double p_dropout, double softmax_scale, bool is_causal,
                                 int64_t wl, int64_t wr) {
    TORCH_CHECK(k_.scalar_type() == at::kFloat && v_.scalar_type() == at::kFloat && !k_.is_cuda() && !v_.is_cuda(),
                "q, k, v must all be fp32 CPU tensors for the host path");
    TORCH_CHECK(p_dropout == 0.0, "the host fp32 path has no dropout");
    TORCH_CHECK(q_.dim() == 4 && k_.dim() == 4 && v_.dim() == 4, "q, k, v must be 4-dimensional");
    const Tensor q = q_.contiguous(), k = k_.contiguous(), v = v_.contiguous();
    const int64_t b = q.size(0), sq = q.size(1), h = q.size(2), d = q.size(3);
    const int64_t sk = k.size(1), hk = k.size(2);
    TORCH_CHECK(b > 0, "batch size must be postive");
    TORCH_CHECK(hk > 0 && h % hk == 0, "Number of heads in key/value must divide number of heads in query");
    expect_shape(k, "k", {b, sk, hk, d});
    expect_shape(v, "v", {b, sk, hk, d});
    if (wl >= sk) wl = -1;                       // flash_api.cpp:289-294
    if (wr >= sk) wr = -1;
    if (is_causal) wr = 0;
    const float *slopes = nullptr;
    int64_t slope_bs = 0;
    Tensor sl;
    if (alibi_slopes_.has_value()) {
        sl = alibi_slopes_.value().contiguous();
        TORCH_CHECK(sl.scalar_type() == at::kFloat && !sl.is_cuda(), "ALiBi slopes must be fp32 on the CPU here");
        TORCH_CHECK(sl.sizes() == at::IntArrayRef({h}) || sl.sizes() == at::IntArrayRef({b, h}),
                    "alibi_slopes must have shape (num_heads) or (batch_size, num_heads)");
        slopes = sl.data_ptr<float>();
        slope_bs = sl.dim() == 2 ? h : 0;
    }
    Tensor out = torch::empty_like(q);
    Tensor lse = torch::empty({b, h, sq}, q.options());
    const float *Q = q.data_ptr<float>(), *K = k.data_ptr<float>(), *V = v.data_ptr<float>();
    float *O = out.data_ptr<float>(), *L = lse.data_ptr<float>();
    const float scale = (float)softmax_scale;
    const int64_t g = h / hk;
    at::parallel_for(0, b * h, 1, [&](int64_t lo, int64_t hi) {
        std::vector<float> s((size_t)sk);
        std::vector<double> acc((size_t)d);
        for (int64_t bh = lo; bh < hi; bh++) {
            const int64_t bi = bh / h, hi_ = bh % h, kh = hi_ / g;
            const float slope = slopes ? slopes[bi * slope_bs + hi_] : 0.f;
            for (int64_t i = 0; i < sq; i++) {
                const float *qr = Q + ((bi * sq + i) * h + hi_) * d;
                const int64_t j_lo = wl < 0 ? 0 : std::max<int64_t>(0, i + sk - sq - wl);
                const int64_t j_hi = wr < 0 ? sk : std::min<int64_t>(sk, i + 1 + sk - sq + wr);
                float m = -std::numeric_limits<float>::infinity();
                for (int64_t j = j_lo; j < j_hi; j++) {
                    const float *kr = K + ((bi * sk + j) * hk + kh) * d;
                    float dot = 0.f;
                    for (int64_t c = 0; c < d; c++) dot += qr[c] * kr[c];
                    dot = dot * scale - slope * std::fabs((float)(i + sk - sq - j));
                    s[(size_t)j] = dot;
                    m = std::max(m, dot);
                }
                float *orow = O + ((bi * sq + i) * h + hi_) * d;
                if (j_hi <= j_lo) {  // fully masked row (softmax.h:178-179)
                    std::fill(orow, orow + d, 0.f);
                    L[bh * sq + i] = std::numeric_limits<float>::infinity();
                    continue;
                }
                double sum = 0.0;
                std::fill(acc.begin(), acc.end(), 0.0);
                for (int64_t j = j_lo; j < j_hi; j++) {
                    const float pj = std::exp(s[(size_t)j] - m);
                    sum += pj;
                    const float *vr = V + ((bi * sk + j) * hk + kh) * d;
                    for (int64_t c = 0; c < d; c++) acc[(size_t)c] += (double)pj * vr[c];
                }
                for (int64_t c = 0; c < d; c++) orow[c] = (float)(acc[(size_t)c] / sum);
                L[bh * sq + i] = m + (float)std::log(sum);
            }
        }
    });
    if (out_.has_value()) {
        TORCH_CHECK(out_.value().sizes() == out.sizes() && out_.value().scalar_type() == at::kFloat,
                    "out must have the shape and dtype of q");
        out_.value().copy_(out);
        out = out_.value();
    }
    Tensor rng_state = torch::zeros({2}, q.options().dtype(at::kLong));
    return {out, q, k, v, out, lse, torch::empty({0}, q.options()), rng_state};
}

// ---------------------------------------------------------------------------------------------
// fwd  (flash_api.cpp:240-420)
std::vector<Tensor> fwd(Tensor &q, const Tensor &k, const Tensor &v, OptTensor &out_,
                        OptTensor &alibi_slopes_, const double p_dropout, const double softmax_scale,
                        bool is_causal, int64_t window_size_left, int64_t window_size_right,
                        const bool return_softmax, std::optional<at::Generator> gen_) {
    // (fp32 CPU tensors are rejected here like any other non-CUDA input, as the reference does
    //  (flash_api.cpp:263-271); the host evaluation is the separate entry `fwd_cpu_fp32`)
    const int dt = dtype_code(q);
    TORCH_CHECK(k.scalar_type() == q.scalar_type(), "query and key must have the same dtype");
    TORCH_CHECK(v.scalar_type() == q.scalar_type(), "query and value must have the same dtype");
    expect_gpu(q, "q"); expect_gpu(k, "k"); expect_gpu(v, "v");
    TORCH_CHECK(q.stride(-1) == 1 && k.stride(-1) == 1 && v.stride(-1) == 1,
                "Input tensor must have contiguous last dimension");
    TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.dim() == 4, "q, k, v must be 4-dimensional");
    const auto in_sizes = q.sizes().vec();
    const int64_t b = in_sizes[0], d_og = in_sizes[3];
    int64_t sq = in_sizes[1], h = in_sizes[2];
    const int64_t sk = k.size(1), hk = k.size(2);
    TORCH_CHECK(b > 0, "batch size must be postive");
    TORCH_CHECK(d_og <= 256, "FlashAttention forward only supports head dimension at most 256");
    TORCH_CHECK(hk > 0 && h % hk == 0, "Number of heads in key/value must divide number of heads in query");

    // Single-token GQA decode: fold the query heads of a KV group into the row dimension
    // (flash_api.cpp:296-304) so one K/V stream serves the whole group.
    int64_t wl = window_size_left, wr = window_size_right;
    if (wl >= sk) wl = -1;
    if (wr >= sk) wr = -1;
    const bool causal_eff = is_causal && !(sq == 1 && !alibi_slopes_.has_value());
    const bool fold = sq == 1 && h > hk && wl < 0 && wr < 0 && !causal_eff && d_og % 8 == 0 &&
                      !alibi_slopes_.has_value() && p_dropout == 0.0;
    const int64_t g = h / hk;
    if (fold) {
        q = q.reshape({b, hk, g, d_og}).transpose(1, 2);
        sq = g;
        h = hk;
    }
    expect_shape(q, "q", {b, sq, h, d_og});
    expect_shape(k, "k", {b, sk, hk, d_og});
    expect_shape(v, "v", {b, sk, hk, d_og});

    Tensor qp = pad_head(q, d_og), kp = pad_head(k, d_og), vp = pad_head(v, d_og);
    Tensor out;
    if (out_.has_value()) {
        out = out_.value();
        TORCH_CHECK(out.scalar_type() == q.scalar_type(), "Output must have the same dtype as inputs");
        expect_gpu(out, "out");
        TORCH_CHECK(out.stride(-1) == 1, "Output tensor must have contiguous last dimension");
        expect_shape(out, "out", {b, in_sizes[1], in_sizes[2], d_og});
        if (fold) out = out.reshape({b, hk, g, d_og}).transpose(1, 2);
        if (d_og % 8 != 0) out = torch::empty_like(qp);
    } else {
        out = torch::empty_like(qp);
    }
    const int64_t d = qp.size(3);

    at::hip::HIPGuardMasqueradingAsCUDA device_guard{(c10::DeviceIndex)q.get_device()};
    auto opts = q.options();
    Tensor lse = torch::empty({b, h, sq}, opts.dtype(at::kFloat));
    Tensor rng_state = torch::empty({2}, opts.dtype(at::kLong));

    vfa_params P{};
    Tensor p_unused = plan_dropout(P, p_dropout, return_softmax, b, h, sq, sk, opts, rng_state, gen_);
    P.dtype = dt;
    set_qkvo(P, qp, kp, vp, out, true);
    P.softmax_lse = lse.data_ptr<float>();
    P.b = b; P.seqlen_q = sq; P.seqlen_k = sk; P.h = h; P.h_k = hk; P.d = d;
    P.softmax_scale = (float)softmax_scale;
    // after folding the call is never causal/local, so the raw flags can be passed through
    P.is_causal = is_causal; P.window_size_left = window_size_left; P.window_size_right = window_size_right;
    if (fold) { P.is_causal = 0; P.window_size_left = -1; P.window_size_right = -1; }
    set_alibi(P, alibi_slopes_, b, h);
    SplitScratch scratch;
    if (sk > 0) {
        if (p_dropout == 0.0) plan_splits(P, scratch, 0, b, h, d, sk, sq, opts);  // flash_api.cpp:206
        P.stream = current_stream();
        run(vfa_mha_fwd(&P));
    } else {  // flash_api.cpp:401-405
        out.zero_();
        lse.fill_(std::numeric_limits<float>::infinity());
    }

    Tensor out_padded = out;
    if (d_og % 8 != 0) {
        out = out.index({"...", torch::indexing::Slice(torch::indexing::None, d_og)});
        if (out_.has_value()) out_.value().copy_(out);
    }
    if (fold) {
        out = out.transpose(1, 2).reshape({b, 1, hk * sq, d_og});
        out_padded = out_padded.transpose(1, 2).reshape({b, 1, hk * sq, d_og});
        qp = qp.transpose(1, 2).reshape({b, 1, hk * sq, d_og});
        lse = lse.reshape({b, hk * sq, 1});
    }
    return {out, qp, kp, vp, out_padded, lse, p_unused, rng_state};
}

// ---------------------------------------------------------------------------------------------
// varlen_fwd  (flash_api.cpp:422-669)
std::vector<Tensor> varlen_fwd(Tensor &q, const Tensor &k, const Tensor &v, OptTensor &out_,
                               const Tensor &cu_seqlens_q, const Tensor &cu_seqlens_k,
                               OptTensor &seqused_k, OptTensor &block_table_, OptTensor &alibi_slopes_,
                               int64_t max_seqlen_q, const int64_t max_seqlen_k, const double p_dropout,
                               const double softmax_scale, const bool zero_tensors, bool is_causal,
                               int64_t window_size_left, int64_t window_size_right,
                               const bool return_softmax, std::optional<at::Generator> gen_) {
    const int dt = dtype_code(q);
    TORCH_CHECK(k.scalar_type() == q.scalar_type(), "query and key must have the same dtype");
    TORCH_CHECK(v.scalar_type() == q.scalar_type(), "query and value must have the same dtype");
    TORCH_CHECK(cu_seqlens_q.scalar_type() == at::kInt, "cu_seqlens_q must have dtype int32");
    TORCH_CHECK(cu_seqlens_k.scalar_type() == at::kInt, "cu_seqlens_k must have dtype int32");
    expect_gpu(q, "q"); expect_gpu(k, "k"); expect_gpu(v, "v");
    expect_gpu(cu_seqlens_q, "cu_seqlens_q"); expect_gpu(cu_seqlens_k, "cu_seqlens_k");
    const bool paged = block_table_.has_value();
    Tensor block_table;
    if (paged) {
        block_table = block_table_.value();
        expect_gpu(block_table, "block_table");
        TORCH_CHECK(block_table.scalar_type() == at::kInt, "block_table must have dtype torch.int32");
        TORCH_CHECK(block_table.stride(-1) == 1, "block_table must have contiguous last dimension");
    }
    TORCH_CHECK(q.stride(-1) == 1 && k.stride(-1) == 1 && v.stride(-1) == 1,
                "Input tensor must have contiguous last dimension");
    expect_contig(cu_seqlens_q, "cu_seqlens_q"); expect_contig(cu_seqlens_k, "cu_seqlens_k");
    TORCH_CHECK(q.dim() == 3, "q must be (total_q, num_heads, head_size)");

    const auto in_sizes = q.sizes().vec();
    const int64_t b = cu_seqlens_q.numel() - 1;
    int64_t h = in_sizes[1];
    const int64_t d_og = in_sizes[2];
    const int64_t hk = paged ? k.size(2) : k.size(1);
    const int64_t max_blocks = paged ? block_table.size(1) : 0;
    const int64_t num_blocks = paged ? k.size(0) : 0;
    const int64_t page = paged ? k.size(1) : 1;
    TORCH_CHECK(!paged || page % 16 == 0, "Paged KV cache block size must be divisible by 16");
    TORCH_CHECK(!(paged && p_dropout > 0.0), "dropout is not supported with a paged KV cache");

    int64_t wl = window_size_left, wr = window_size_right;
    if (wl >= max_seqlen_k) wl = -1;
    if (wr >= max_seqlen_k) wr = -1;
    const bool causal_eff = is_causal && !(max_seqlen_q == 1 && !alibi_slopes_.has_value());
    const bool fold = max_seqlen_q == 1 && h > hk && wl < 0 && wr < 0 && !causal_eff &&
                      d_og % 8 == 0 && !alibi_slopes_.has_value() && p_dropout == 0.0;
    const int64_t g = h / std::max<int64_t>(hk, 1);
    bool dense_q = false;
    if (fold) {  // flash_api.cpp:500-507: every sequence has exactly one query token
        q = q.reshape({b, hk, g, d_og}).transpose(1, 2).reshape({b * g, hk, d_og});
        max_seqlen_q = g;
        h = hk;
        dense_q = true;
    }
    const int64_t total_q = q.size(0);
    TORCH_CHECK(b > 0, "batch size must be positive");
    TORCH_CHECK(d_og <= 256, "FlashAttention forward only supports head dimension at most 256");
    TORCH_CHECK(hk > 0 && h % hk == 0, "Number of heads in key/value must divide number of heads in query");
    expect_shape(q, "q", {total_q, h, d_og});
    if (!paged) {
        const int64_t total_k = k.size(0);
        expect_shape(k, "k", {total_k, hk, d_og});
        expect_shape(v, "v", {total_k, hk, d_og});
    } else {
        expect_shape(k, "k", {num_blocks, page, hk, d_og});
        expect_shape(v, "v", {num_blocks, page, hk, d_og});
        expect_shape(block_table, "block_table", {b, max_blocks});
    }
    expect_shape(cu_seqlens_q, "cu_seqlens_q", {b + 1});
    expect_shape(cu_seqlens_k, "cu_seqlens_k", {b + 1});
    if (seqused_k.has_value()) {
        const Tensor &s = seqused_k.value();
        TORCH_CHECK(s.scalar_type() == at::kInt, "seqused_k must have dtype int32");
        TORCH_CHECK(s.is_cuda(), "seqused_k must be on CUDA device");
        TORCH_CHECK(s.is_contiguous(), "seqused_k must be contiguous");
        expect_shape(s, "seqused_k", {b});
    }

    Tensor qp = pad_head(q, d_og), kp = pad_head(k, d_og), vp = pad_head(v, d_og);
    Tensor out;
    if (out_.has_value()) {
        out = out_.value();
        TORCH_CHECK(out.scalar_type() == q.scalar_type(), "Output must have the same dtype as inputs");
        expect_gpu(out, "out");
        TORCH_CHECK(out.stride(-1) == 1, "Output tensor must have contiguous last dimension");
        expect_shape(out, "out", {in_sizes[0], in_sizes[1], d_og});
        if (fold) out = out.reshape({b, hk, g, d_og}).transpose(1, 2).reshape({b * g, hk, d_og});
        if (d_og % 8 != 0) out = torch::empty_like(qp);
    } else {
        out = torch::empty_like(qp);
    }
    const int64_t d = qp.size(2);

    at::hip::HIPGuardMasqueradingAsCUDA device_guard{(c10::DeviceIndex)q.get_device()};
    auto opts = q.options();
    Tensor lse = torch::empty({b, h, max_seqlen_q}, opts.dtype(at::kFloat));
    Tensor rng_state = torch::empty({2}, opts.dtype(at::kLong));
    if (zero_tensors) {
        out.zero_();
        lse.fill_(-std::numeric_limits<float>::infinity());
    }

    vfa_params P{};
    Tensor p_unused = plan_dropout(P, p_dropout, return_softmax, b, h, max_seqlen_q, max_seqlen_k, opts,
                                   rng_state, gen_);
    P.dtype = dt;
    set_qkvo(P, qp, kp, vp, out, false);
    P.softmax_lse = lse.data_ptr<float>();
    P.b = b; P.seqlen_q = max_seqlen_q; P.seqlen_k = max_seqlen_k; P.h = h; P.h_k = hk; P.d = d;
    P.total_q = (int)total_q;
    P.cu_seqlens_k = cu_seqlens_k.data_ptr<int32_t>();
    P.is_seqlens_k_cumulative = 1;
    if (seqused_k.has_value()) P.seqused_k = seqused_k.value().data_ptr<int32_t>();
    if (paged) {
        P.block_table = block_table.data_ptr<int32_t>();
        P.block_table_batch_stride = block_table.stride(0);
        P.k_batch_stride = kp.stride(0);
        P.v_batch_stride = vp.stride(0);
        P.page_block_size = page;
    }
    P.softmax_scale = (float)softmax_scale;
    P.is_causal = is_causal; P.window_size_left = window_size_left; P.window_size_right = window_size_right;
    set_alibi(P, alibi_slopes_, b, h);
    P.stream = current_stream();
    SplitScratch scratch;
    if (max_seqlen_k > 0) {
        if (dense_q) {
            // one query token per sequence, heads folded into rows: a dense (b, g, hk, d) query
            // against the ragged keys; this is the decode path (split-KV allowed)
            P.is_causal = 0; P.window_size_left = -1; P.window_size_right = -1;
            P.q_batch_stride = g * qp.stride(0); P.o_batch_stride = g * out.stride(0);
            plan_splits(P, scratch, 0, b, h, d, max_seqlen_k, max_seqlen_q, opts);
            P.cu_seqlens_q = nullptr;
            // ragged keys with a dense query go through the kv-cache entry semantics
            run(vfa_mha_varlen_fwd(&P));
        } else {
            P.cu_seqlens_q = cu_seqlens_q.data_ptr<int32_t>();
            run(vfa_mha_varlen_fwd(&P));
        }
    } else {
        out.zero_();
        lse.fill_(std::numeric_limits<float>::infinity());
    }

    Tensor out_padded = out;
    if (d_og % 8 != 0) {
        out = out.index({"...", torch::indexing::Slice(torch::indexing::None, d_og)});
        if (out_.has_value()) out_.value().copy_(out);
    }
    if (fold) {
        std::vector<int64_t> before = {b, max_seqlen_q, hk, d_og}, after = {b, hk * max_seqlen_q, d_og};
        out = out.reshape(before).transpose(1, 2).reshape(after);
        out_padded = out_padded.reshape(before).transpose(1, 2).reshape(after);
        qp = qp.reshape(before).transpose(1, 2).reshape(after);
        lse = lse.reshape({b, hk * max_seqlen_q, 1});
    }
    return {out, qp, kp, vp, out_padded, lse, p_unused, rng_state};
}

// ---------------------------------------------------------------------------------------------
// fwd_kvcache  (flash_api.cpp:671-937)
std::vector<Tensor> fwd_kvcache(Tensor &q, const Tensor &kcache, const Tensor &vcache,
                                std::optional<const Tensor> &k_, std::optional<const Tensor> &v_,
                                std::optional<const Tensor> &seqlens_k_,
                                std::optional<const Tensor> &rotary_cos_,
                                std::optional<const Tensor> &rotary_sin_,
                                std::optional<const Tensor> &cache_batch_idx_, OptTensor &block_table_,
                                OptTensor &alibi_slopes_, OptTensor &out_, const double softmax_scale,
                                bool is_causal, int64_t window_size_left, int64_t window_size_right,
                                bool is_rotary_interleaved, int64_t num_splits) {
    const int dt = dtype_code(q);
    TORCH_CHECK(kcache.scalar_type() == q.scalar_type(), "query and key must have the same dtype");
    TORCH_CHECK(vcache.scalar_type() == q.scalar_type(), "query and value must have the same dtype");
    expect_gpu(q, "q"); expect_gpu(kcache, "kcache"); expect_gpu(vcache, "vcache");
    TORCH_CHECK(q.stride(-1) == 1 && kcache.stride(-1) == 1 && vcache.stride(-1) == 1,
                "Input tensor must have contiguous last dimension");
    const bool paged = block_table_.has_value();
    Tensor block_table;
    if (paged) {
        TORCH_CHECK(!cache_batch_idx_.has_value(), "Paged KVcache does not support cache_batch_idx");
        block_table = block_table_.value();
        expect_gpu(block_table, "block_table");
        TORCH_CHECK(block_table.scalar_type() == at::kInt, "block_table must have dtype torch.int32");
        TORCH_CHECK(block_table.stride(-1) == 1, "block_table must have contiguous last dimension");
    }
    TORCH_CHECK(q.dim() == 4, "q must be (batch, seqlen_q, num_heads, head_size)");
    const auto in_sizes = q.sizes().vec();
    const int64_t b = in_sizes[0], d_og = in_sizes[3];
    int64_t sq = in_sizes[1], h = in_sizes[2];
    const int64_t max_blocks = paged ? block_table.size(1) : 0;
    const int64_t num_blocks = paged ? kcache.size(0) : 0;
    const int64_t page = paged ? kcache.size(1) : 1;
    TORCH_CHECK(!paged || page % 16 == 0, "Paged KV cache block size must be divisible by 16");
    const int64_t sk = paged ? max_blocks * page : kcache.size(1);
    const int64_t hk = kcache.size(2);
    const int64_t b_c = paged ? b : kcache.size(0);
    TORCH_CHECK(b > 0, "batch size must be postive");
    TORCH_CHECK(d_og <= 256, "FlashAttention forward only supports head dimension at most 256");
    TORCH_CHECK(hk > 0 && h % hk == 0, "Number of heads in key/value must divide number of heads in query");

    const bool causal_eff = is_causal && !(sq == 1 && !alibi_slopes_.has_value());
    const bool fold = sq == 1 && h > hk && window_size_left < 0 && window_size_right < 0 &&
                      !causal_eff && d_og % 8 == 0 && !alibi_slopes_.has_value();
    const int64_t g = h / hk;
    if (fold) {  // flash_api.cpp:747-755
        q = q.reshape({b, hk, g, d_og}).transpose(1, 2);
        sq = g;
        h = hk;
    }
    expect_shape(q, "q", {b, sq, h, d_og});
    if (!paged) {
        expect_shape(kcache, "kcache", {b_c, sk, hk, d_og});
        expect_shape(vcache, "vcache", {b_c, sk, hk, d_og});
    } else {
        expect_shape(kcache, "kcache", {num_blocks, page, hk, d_og});
        expect_shape(vcache, "vcache", {num_blocks, page, hk, d_og});
        expect_shape(block_table, "block_table", {b, max_blocks});
    }

    Tensor qp = pad_head(q, d_og), kcp = pad_head(kcache, d_og), vcp = pad_head(vcache, d_og);
    Tensor out;
    if (out_.has_value()) {
        out = out_.value();
        TORCH_CHECK(out.scalar_type() == q.scalar_type(), "Output must have the same dtype as inputs");
        expect_gpu(out, "out");
        TORCH_CHECK(out.stride(-1) == 1, "Output tensor must have contiguous last dimension");
        // The reference checks `out` against the FOLDED query shape here (flash_api.cpp:787), so a
        // caller-shaped (b, 1, h, d) output is rejected whenever the GQA decode fold applies, while
        // mha_fwd re-views it (flash_api.cpp:327-330).  Accept both: the caller's shape is viewed
        // like the query, the folded shape is taken as is.
        if (fold && out.dim() == 4 && out.size(1) == 1 && out.size(2) == hk * g) {
            expect_shape(out, "out", {b, 1, hk * g, d_og});
            out = out.reshape({b, hk, g, d_og}).transpose(1, 2);
        } else {
            expect_shape(out, "out", {b, sq, h, d_og});
        }
        if (d_og % 8 != 0) out = torch::empty_like(qp);
    } else {
        out = torch::empty_like(qp);
    }
    const int64_t d = qp.size(3);

    at::hip::HIPGuardMasqueradingAsCUDA device_guard{(c10::DeviceIndex)q.get_device()};
    auto opts = q.options();
    Tensor lse = torch::empty({b, h, sq}, opts.dtype(at::kFloat));

    vfa_params P{};
    P.dtype = dt;
    set_qkvo(P, qp, kcp, vcp, out, true);
    P.softmax_lse = lse.data_ptr<float>();
    P.b = b; P.seqlen_q = sq; P.seqlen_k = sk; P.h = h; P.h_k = hk; P.d = d;
    P.softmax_scale = (float)softmax_scale;
    P.is_causal = is_causal; P.window_size_left = window_size_left; P.window_size_right = window_size_right;
    if (fold) { P.is_causal = 0; P.window_size_left = -1; P.window_size_right = -1; }

    Tensor knew, vnew;
    if (k_.has_value()) {
        TORCH_CHECK(v_.has_value(), "If key is supplied, value must also be passed in");
        TORCH_CHECK(seqlens_k_.has_value(), "If key is supplied, seqlens_k must also be passed in");
        TORCH_CHECK(sq <= sk, "If key is supplied, it must have seqlen <= the seqlen of the KV cache");
        const Tensor &kk = k_.value(), &vv = v_.value();
        TORCH_CHECK(kk.scalar_type() == q.scalar_type(), "Key must have the same dtype as query");
        TORCH_CHECK(vv.scalar_type() == q.scalar_type(), "Value must have the same dtype as query");
        expect_gpu(kk, "k"); expect_gpu(vv, "v");
        TORCH_CHECK(kk.stride(-1) == 1, "Key tensor must have contiguous last dimension");
        TORCH_CHECK(vv.stride(-1) == 1, "Value tensor must have contiguous last dimension");
        const int64_t s_new = kk.size(1);
        expect_shape(kk, "k", {b, s_new, hk, d_og});
        expect_shape(vv, "v", {b, s_new, hk, d_og});
        knew = pad_head(kk, d_og); vnew = pad_head(vv, d_og);
        P.knew = knew.data_ptr(); P.vnew = vnew.data_ptr(); P.seqlen_knew = s_new;
        P.knew_batch_stride = knew.stride(0); P.knew_row_stride = knew.stride(-3); P.knew_head_stride = knew.stride(-2);
        P.vnew_batch_stride = vnew.stride(0); P.vnew_row_stride = vnew.stride(-3); P.vnew_head_stride = vnew.stride(-2);
    }
    if (seqlens_k_.has_value()) {
        const Tensor &s = seqlens_k_.value();
        TORCH_CHECK(s.scalar_type() == at::kInt, "seqlens_k must have dtype int32");
        expect_gpu(s, "seqlens_k"); expect_contig(s, "seqlens_k");
        expect_shape(s, "seqlens_k", {b});
        P.cu_seqlens_k = s.data_ptr<int32_t>();
    }
    P.is_seqlens_k_cumulative = !seqlens_k_.has_value();
    if (rotary_cos_.has_value()) {
        TORCH_CHECK(k_.has_value(), "If rotary cos/sin are provided, new key / value to be appended to KV cache must also be provided");
        const Tensor &rc = rotary_cos_.value();
        expect_gpu(rc, "rotary_cos");
        P.rotary_dim = rc.size(1) * 2;
        TORCH_CHECK(P.rotary_dim <= d, "rotary_dim must be <= headdim");
        TORCH_CHECK(P.rotary_dim % 16 == 0, "Only rotary dimensions divisible by 16 are currently supported");
        const int64_t s_ro = rc.size(0);
        TORCH_CHECK(s_ro >= sk, "cos/sin seqlen must be at least the seqlen of KV cache");
        expect_shape(rc, "rotary_cos", {s_ro, P.rotary_dim / 2});
        expect_contig(rc, "rotary_cos");
        TORCH_CHECK(rc.scalar_type() == q.scalar_type(), "rotary_cos must have the same dtype as query");
        TORCH_CHECK(rotary_sin_.has_value(), "If rotary cos is provided, rotary sin must also be provided");
        const Tensor &rs = rotary_sin_.value();
        expect_gpu(rs, "rotary_sin");
        expect_shape(rs, "rotary_sin", {s_ro, P.rotary_dim / 2});
        expect_contig(rs, "rotary_sin");
        TORCH_CHECK(rs.scalar_type() == q.scalar_type(), "rotary_cos must have the same dtype as query");
        P.rotary_cos = rc.data_ptr(); P.rotary_sin = rs.data_ptr();
        P.is_rotary_interleaved = is_rotary_interleaved;
        P.seqlen_ro = (int)s_ro;
    }
    if (cache_batch_idx_.has_value()) {
        const Tensor &c = cache_batch_idx_.value();
        expect_gpu(c, "cache_batch_idx"); expect_contig(c, "cache_batch_idx");
        TORCH_CHECK(c.scalar_type() == at::kInt, "cache_batch_idx must have dtype int32");
        P.cache_batch_idx = c.data_ptr<int32_t>();
    }
    if (paged) {
        P.block_table = block_table.data_ptr<int32_t>();
        P.block_table_batch_stride = block_table.stride(0);
        P.page_block_size = page;
    }
    set_alibi(P, alibi_slopes_, b, h);
    SplitScratch scratch;
    plan_splits(P, scratch, (int)num_splits, b, h, d, sk, sq, opts);
    P.stream = current_stream();
    run(vfa_mha_fwd_kvcache(&P));

    if (d_og % 8 != 0) {
        out = out.index({"...", torch::indexing::Slice(torch::indexing::None, d_og)});
        if (out_.has_value()) out_.value().copy_(out);
        if (k_.has_value()) {  // flash_api.cpp:924-929
            kcache.copy_(kcp.index({"...", torch::indexing::Slice(torch::indexing::None, d_og)}));
            vcache.copy_(vcp.index({"...", torch::indexing::Slice(torch::indexing::None, d_og)}));
        }
    }
    if (fold) {
        out = out.transpose(1, 2).reshape({b, 1, hk * sq, d_og});
        lse = lse.reshape({b, hk * sq, 1});
    }
    return {out, lse};
}

// ---------------------------------------------------------------------------------------------
// bwd / varlen_bwd  (call sites flash_attn_interface.py:135-154, :183-207; Flash_bwd_params,
// csrc/flash_attn/src/flash.h:145-183)
std::vector<Tensor> bwd_impl(const Tensor &dout_in, const Tensor &q_in, const Tensor &k_in, const Tensor &v_in,
                             const Tensor &out_in, const Tensor &softmax_lse, OptTensor &dq_, OptTensor &dk_,
                             OptTensor &dv_, const Tensor *cu_seqlens_q, const Tensor *cu_seqlens_k,
                             OptTensor &alibi_slopes_, int64_t max_seqlen_q, int64_t max_seqlen_k,
                             const double p_dropout, const double softmax_scale, const bool zero_tensors,
                             const bool is_causal, int64_t window_size_left, int64_t window_size_right,
                             const bool deterministic, std::optional<at::Generator> gen_,
                             OptTensor &rng_state_) {
    const bool varlen = cu_seqlens_q != nullptr;
    const int dt = dtype_code(q_in);
    for (const Tensor *t : {&k_in, &v_in, &out_in, &dout_in})
        TORCH_CHECK(t->scalar_type() == q_in.scalar_type(), "query, key, value, out and dout must have the same dtype");
    for (const Tensor *t : {&q_in, &k_in, &v_in, &out_in, &dout_in, &softmax_lse}) expect_gpu(*t, "input");
    for (const Tensor *t : {&q_in, &k_in, &v_in, &out_in, &dout_in})
        TORCH_CHECK(t->stride(-1) == 1, "Input tensor must have contiguous last dimension");
    // Head dims that are not a multiple of 8: the forward returned q, k, v, out zero-padded to the
    // next multiple (flash_api.cpp:310-319) and autograd saved those, while dout arrives with the
    // caller's head dim; upstream's mha_bwd pads dout (and slices dq, dk, dv back at the end).
    // Unpadded q / k / v / out are accepted too and padded here the same way.
    const int64_t d_og = dout_in.size(-1);
    const int64_t d_in = q_in.size(-1);
    const int64_t d_pad = (std::max(d_og, d_in) + 7) / 8 * 8;
    TORCH_CHECK(d_og <= d_in && d_pad - d_og < 8 && (d_in == d_og || d_in == d_pad),
                "dout must have the head size of q, or the head size q had before it was padded to a multiple of 8");
    auto pad_to = [&](const Tensor &t) {
        TORCH_CHECK(t.size(-1) == d_og || t.size(-1) == d_pad, "q, k, v, out must share one head size");
        if (t.size(-1) == d_pad) return t;
        namespace F = torch::nn::functional;
        return F::pad(t, F::PadFuncOptions({0, d_pad - t.size(-1)}));
    };
    const Tensor q = pad_to(q_in), k = pad_to(k_in), v = pad_to(v_in), out = pad_to(out_in), dout = pad_to(dout_in);
    TORCH_CHECK(p_dropout >= 0.0 && p_dropout < 1.0, "p_dropout must be in [0, 1)");
    TORCH_CHECK(softmax_lse.scalar_type() == at::kFloat, "softmax_lse must be fp32");

    int64_t b, sq, sk, h, hk, d, total_q = 0, total_k = 0;
    if (varlen) {
        TORCH_CHECK(cu_seqlens_q->scalar_type() == at::kInt && cu_seqlens_k->scalar_type() == at::kInt,
                    "cu_seqlens must have dtype int32");
        expect_gpu(*cu_seqlens_q, "cu_seqlens_q"); expect_gpu(*cu_seqlens_k, "cu_seqlens_k");
        expect_contig(*cu_seqlens_q, "cu_seqlens_q"); expect_contig(*cu_seqlens_k, "cu_seqlens_k");
        b = cu_seqlens_q->numel() - 1;
        total_q = q.size(0); h = q.size(1); d = q.size(2);
        total_k = k.size(0); hk = k.size(1);
        sq = max_seqlen_q; sk = max_seqlen_k;
        expect_shape(k, "k", {total_k, hk, d}); expect_shape(v, "v", {total_k, hk, d});
        expect_shape(out, "out", {total_q, h, d}); expect_shape(dout, "dout", {total_q, h, d});
    } else {
        b = q.size(0); sq = q.size(1); h = q.size(2); d = q.size(3);
        sk = k.size(1); hk = k.size(2);
        expect_shape(k, "k", {b, sk, hk, d}); expect_shape(v, "v", {b, sk, hk, d});
        expect_shape(out, "out", {b, sq, h, d}); expect_shape(dout, "dout", {b, sq, h, d});
    }
    TORCH_CHECK(b > 0, "batch size must be positive");
    TORCH_CHECK(d % 8 == 0, "head_size should be a multiple of 8");
    TORCH_CHECK(d <= 256, "FlashAttention backward only supports head dimension at most 256");
    TORCH_CHECK(hk > 0 && h % hk == 0, "Number of heads in key/value must divide number of heads in query");

    // Caller-provided gradients have the shape of the q / k / v that were passed in; when those were
    // not padded the kernels write padded temporaries and the head-dim prefix is copied back.
    auto grad_buf = [&](OptTensor &g_, const Tensor &like_in, const Tensor &like_pad) {
        if (!g_.has_value()) return torch::empty_like(like_pad);
        const Tensor &g = g_.value();
        TORCH_CHECK(g.scalar_type() == q.scalar_type(), "dq, dk, dv must have the same dtype as q");
        expect_gpu(g, "gradient");
        TORCH_CHECK(g.stride(-1) == 1, "gradient must have contiguous last dimension");
        TORCH_CHECK(g.sizes() == like_in.sizes(), "dq, dk, dv must have the shapes of q, k, v");
        return g.size(-1) == d_pad ? g : torch::empty_like(like_pad);
    };
    Tensor dq = grad_buf(dq_, q_in, q), dk = grad_buf(dk_, k_in, k), dv = grad_buf(dv_, v_in, v);

    at::hip::HIPGuardMasqueradingAsCUDA device_guard{(c10::DeviceIndex)q.get_device()};
    auto opts = q.options();
    const int64_t sq_rounded = (sq + 127) / 128 * 128;
    Tensor softmax_d = torch::empty({b, h, sq_rounded}, opts.dtype(at::kFloat));
    // No fp32 dq accumulator (the kernels accumulate in registers, flash.h:153-156); fp32 dK / dV
    // slabs only when the heads of a KV group are split over workgroups to fill the chip.
    const int head_splits = vfa_bwd_head_splits((int)b, (int)h, (int)hk, (int)sk, cu_count());
    const int64_t part_rows = varlen ? total_k : b * sk;
    Tensor dk_accum, dv_accum;
    if (head_splits > 1) {
        dk_accum = torch::empty({head_splits, part_rows, hk, d}, opts.dtype(at::kFloat));
        dv_accum = torch::empty({head_splits, part_rows, hk, d}, opts.dtype(at::kFloat));
    }
    if (zero_tensors) { dq.zero_(); dk.zero_(); dv.zero_(); softmax_d.zero_(); }

    vfa_params P{};
    P.dtype = dt;
    set_qkvo(P, q, k, v, out, !varlen);
    P.softmax_lse = softmax_lse.data_ptr<float>();
    P.b = b; P.seqlen_q = sq; P.seqlen_k = sk; P.h = h; P.h_k = hk; P.d = d;
    if (varlen) {
        P.cu_seqlens_q = cu_seqlens_q->data_ptr<int32_t>();
        P.cu_seqlens_k = cu_seqlens_k->data_ptr<int32_t>();
        P.is_seqlens_k_cumulative = 1;
    }
    P.softmax_scale = (float)softmax_scale;
    P.is_causal = is_causal; P.window_size_left = window_size_left; P.window_size_right = window_size_right;
    set_alibi(P, alibi_slopes_, b, h);
    P.dout = dout.data_ptr(); P.dq = dq.data_ptr(); P.dk = dk.data_ptr(); P.dv = dv.data_ptr();
    P.do_row_stride = dout.stride(-3); P.do_head_stride = dout.stride(-2);
    P.dq_row_stride = dq.stride(-3); P.dq_head_stride = dq.stride(-2);
    P.dk_row_stride = dk.stride(-3); P.dk_head_stride = dk.stride(-2);
    P.dv_row_stride = dv.stride(-3); P.dv_head_stride = dv.stride(-2);
    if (!varlen) {
        P.do_batch_stride = dout.stride(0); P.dq_batch_stride = dq.stride(0);
        P.dk_batch_stride = dk.stride(0); P.dv_batch_stride = dv.stride(0);
    }
    P.dsoftmax_sum = softmax_d.data_ptr<float>();
    P.deterministic = deterministic;
    P.stream = current_stream();
    if (head_splits > 1) {
        P.bwd_head_splits = head_splits;
        P.bwd_part_rows = part_rows;
        P.dk_accum = dk_accum.data_ptr<float>();
        P.dv_accum = dv_accum.data_ptr<float>();
    }
    // The forward's Philox state: the rng_state tensor it returned (read on the device), else a
    // fresh draw from the generator like the upstream backward does when none is passed.
    P.p_dropout = (float)p_dropout;
    if (p_dropout > 0.0) {
        if (rng_state_.has_value()) {
            const Tensor &rs = rng_state_.value();
            TORCH_CHECK(rs.is_cuda() && rs.scalar_type() == at::kLong && rs.numel() == 2 && rs.is_contiguous(),
                        "rng_state must be a CUDA int64 tensor with 2 elements");
            P.rng_state = (uint64_t *)rs.data_ptr<int64_t>();
        } else {
            auto gen = at::get_generator_or_default<at::CUDAGeneratorImpl>(
                gen_, at::cuda::detail::getDefaultCUDAGenerator());
            std::lock_guard<std::mutex> lock(gen->mutex_);
            const auto seed_offset = gen->philox_engine_inputs((uint64_t)(b * h * 32));
            P.philox_seed = seed_offset.first;
            P.philox_offset = seed_offset.second;
        }
    }
    if (sk > 0 && sq > 0) {
        run(varlen ? vfa_mha_varlen_bwd(&P) : vfa_mha_bwd(&P));
    } else {
        dq.zero_(); dk.zero_(); dv.zero_(); softmax_d.zero_();
    }
    if (d_og != d_pad) {  // upstream mha_bwd: slice the padding off (and fill the caller's buffers)
        auto unpad = [&](Tensor &g, OptTensor &g_) {
            if (g_.has_value() && g_.value().size(-1) != d_pad) g_.value().copy_(g.narrow(-1, 0, g_.value().size(-1)));
            g = g.narrow(-1, 0, d_og);
        };
        unpad(dq, dq_); unpad(dk, dk_); unpad(dv, dv_);
    }
    return {dq, dk, dv, softmax_d};
}

std::vector<Tensor> bwd(const Tensor &dout, const Tensor &q, const Tensor &k, const Tensor &v,
                        const Tensor &out, const Tensor &softmax_lse, OptTensor &dq_, OptTensor &dk_,
                        OptTensor &dv_, OptTensor &alibi_slopes_, const double p_dropout,
                        const double softmax_scale, const bool is_causal, int64_t window_size_left,
                        int64_t window_size_right, const bool deterministic,
                        std::optional<at::Generator> gen_, OptTensor &rng_state) {
    return bwd_impl(dout, q, k, v, out, softmax_lse, dq_, dk_, dv_, nullptr, nullptr, alibi_slopes_, 0, 0,
                    p_dropout, softmax_scale, false, is_causal, window_size_left, window_size_right,
                    deterministic, gen_, rng_state);
}

std::vector<Tensor> varlen_bwd(const Tensor &dout, const Tensor &q, const Tensor &k, const Tensor &v,
                               const Tensor &out, const Tensor &softmax_lse, OptTensor &dq_, OptTensor &dk_,
                               OptTensor &dv_, const Tensor &cu_seqlens_q, const Tensor &cu_seqlens_k,
                               OptTensor &alibi_slopes_, const int64_t max_seqlen_q,
                               const int64_t max_seqlen_k, const double p_dropout,
                               const double softmax_scale, const bool zero_tensors, const bool is_causal,
                               int64_t window_size_left, int64_t window_size_right,
                               const bool deterministic, std::optional<at::Generator> gen_,
                               OptTensor &rng_state) {
    return bwd_impl(dout, q, k, v, out, softmax_lse, dq_, dk_, dv_, &cu_seqlens_q, &cu_seqlens_k,
                    alibi_slopes_, max_seqlen_q, max_seqlen_k, p_dropout, softmax_scale, zero_tensors,
                    is_causal, window_size_left, window_size_right, deterministic, gen_, rng_state);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "FlashAttention (MI355X / gfx950)";
    m.def("fwd", &fwd, "Forward pass");
    // No counterpart in the reference (which accepts fp16 / bf16 CUDA tensors only): a plain host
    // loop for fp32 CPU tensors, forward only, for the one configuration of the benchmark list that
    // runs without a GPU (batch 2, seqlen 128, 4 heads, head dim 64 against torch SDPA).  The Python
    // package routes to it only when asked (vllm_flash_attn.enable_cpu_fp32).
    m.def("fwd_cpu_fp32",
          [](const Tensor &q, const Tensor &k, const Tensor &v, OptTensor out_, OptTensor alibi_slopes_, double p_dropout,
             double softmax_scale, bool is_causal, int64_t wl, int64_t wr) {
              TORCH_CHECK(!q.is_cuda() && q.scalar_type() == at::kFloat, "fwd_cpu_fp32 takes fp32 CPU tensors");
              return fwd_cpu_fp32(q, k, v, out_, alibi_slopes_, p_dropout, softmax_scale, is_causal, wl, wr);
          },
          "Forward pass on fp32 CPU tensors (host loop, forward only)");
    m.def("varlen_fwd", &varlen_fwd, "Forward pass (variable length)");
    m.def("bwd", &bwd, "Backward pass");
    m.def("varlen_bwd", &varlen_bwd, "Backward pass (variable length)");
    m.def("fwd_kvcache", &fwd_kvcache, "Forward pass, with KV-cache");
}
