"""Python API of the MI355X FlashAttention-2 build.

Same public functions, argument order, defaults and return conventions as the reference's
vllm_flash_attn/flash_attn_interface.py (flash_attn_func :793, flash_attn_kvpacked_func :716,
flash_attn_qkvpacked_func :658, flash_attn_varlen_func :1026, flash_attn_varlen_kvpacked_func :935,
flash_attn_varlen_qkvpacked_func :870, flash_attn_with_kvcache :1119), on top of the extension module
`vllm_flash_attn_2_cuda` built from this repository for gfx950.

Deliberate differences from the reference wrappers:
  * every autograd `backward` returns one gradient slot per `forward` input (the reference's return
    one too few since the `out` argument was added, so its backward cannot run), and the packed
    variants take `out` positionally (the reference declares it keyword-only but passes it
    positionally through `.apply`, a TypeError);
  * the extension exports `bwd` / `varlen_bwd` (the reference fork is forward-only);
  * dropout_p > 0 and return_attn_probs work (the fork compiles dropout out, setup.py:204): the
    keep mask is the reference kernel's Philox pattern (dropout.h, philox.cuh) for the state drawn
    from torch's CUDA generator.
  * opt-in only: after `enable_cpu_fp32(True)` (or with VLLM_FLASH_ATTN_CPU_FP32=1 in the
    environment) `flash_attn_func` evaluates fp32 CPU tensors with the extension's host loop, forward
    only (inputs that require grad are refused) -- the benchmark list's CPU configuration.  Without
    the opt-in CPU tensors raise the reference's "must be on CUDA" error.
There is no CPU or eager fallback for the kernels: without the HIP extension the import fails.
"""
import os
from typing import Optional, Union

import torch

# The HIP kernels are imported after torch, like the reference imports its CUDA module (:10).
import vllm_flash_attn_2_cuda as flash_attn_cuda


def _get_block_size_n(device, head_dim, is_dropout, is_causal):
    """Key-tile width of the forward kernel (the reference's helper of the same name, :15-43, kept
    importable for test code, which needs it to re-normalise S_dmask).  The gfx950 prefill kernel
    always walks 64-key tiles."""
    assert head_dim <= 256
    return 64


def _last_dim_contiguous(x):
    return x.contiguous() if x is not None and x.stride(-1) != 1 else x


_cpu_fp32 = os.environ.get("VLLM_FLASH_ATTN_CPU_FP32", "0") == "1"


def enable_cpu_fp32(on=True):
    """Opt in to (or out of) the host evaluation of fp32 CPU tensors in `flash_attn_func`
    (forward only; no counterpart in the reference, whose extension rejects them).  Returns the
    previous setting."""
    global _cpu_fp32
    prev, _cpu_fp32 = _cpu_fp32, bool(on)
    return prev


def _flash_attn_forward(q, k, v, dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                        return_softmax, *, out=None):
    q, k, v = (_last_dim_contiguous(x) for x in (q, k, v))
    if _cpu_fp32 and not q.is_cuda and q.dtype == torch.float32:
        return tuple(flash_attn_cuda.fwd_cpu_fp32(q, k, v, out, alibi_slopes, dropout_p, softmax_scale, causal,
                                                  window_size[0], window_size[1]))
    out, q, k, v, out_padded, softmax_lse, S_dmask, rng_state = flash_attn_cuda.fwd(
        q, k, v, out, alibi_slopes, dropout_p, softmax_scale, causal, window_size[0], window_size[1],
        return_softmax, None)
    return out, q, k, v, out_padded, softmax_lse, S_dmask, rng_state


def _flash_attn_varlen_forward(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                               dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                               return_softmax, block_table, *, out=None):
    q, k, v = (_last_dim_contiguous(x) for x in (q, k, v))
    out, q, k, v, out_padded, softmax_lse, S_dmask, rng_state = flash_attn_cuda.varlen_fwd(
        q, k, v, out, cu_seqlens_q, cu_seqlens_k, None, block_table, alibi_slopes, max_seqlen_q,
        max_seqlen_k, dropout_p, softmax_scale, False, causal, window_size[0], window_size[1],
        return_softmax, None)
    return out, q, k, v, out_padded, softmax_lse, S_dmask, rng_state


def _flash_attn_backward(dout, q, k, v, out, softmax_lse, dq, dk, dv, dropout_p, softmax_scale,
                         causal, window_size, alibi_slopes, deterministic, rng_state=None):
    dout, q, k, v, out = (_last_dim_contiguous(x) for x in (dout, q, k, v, out))
    dq, dk, dv, softmax_d = flash_attn_cuda.bwd(
        dout, q, k, v, out, softmax_lse, dq, dk, dv, alibi_slopes, dropout_p, softmax_scale, causal,
        window_size[0], window_size[1], deterministic, None, rng_state)
    return dq, dk, dv, softmax_d


def _flash_attn_varlen_backward(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q,
                                cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale,
                                causal, window_size, alibi_slopes, deterministic, rng_state=None):
    dout, q, k, v, out = (_last_dim_contiguous(x) for x in (dout, q, k, v, out))
    dq, dk, dv, softmax_d = flash_attn_cuda.varlen_bwd(
        dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k, alibi_slopes,
        max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, False, causal, window_size[0],
        window_size[1], deterministic, None, rng_state)
    return dq, dk, dv, softmax_d


def _save_cfg(ctx, dropout_p, softmax_scale, causal, window_size, alibi_slopes, deterministic):
    ctx.dropout_p = dropout_p
    ctx.softmax_scale = softmax_scale
    ctx.causal = causal
    ctx.window_size = window_size
    ctx.alibi_slopes = alibi_slopes
    ctx.deterministic = deterministic


def _dense_fwd(ctx, q, k, v, dropout_p, softmax_scale, causal, window_size, alibi_slopes,
               deterministic, return_softmax, out):
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    out, q, k, v, out_padded, softmax_lse, S_dmask, rng_state = _flash_attn_forward(
        q, k, v, dropout_p, softmax_scale, causal=causal, window_size=window_size,
        alibi_slopes=alibi_slopes, return_softmax=return_softmax and dropout_p > 0, out=out)
    ctx.save_for_backward(q, k, v, out_padded, softmax_lse, rng_state)
    _save_cfg(ctx, dropout_p, softmax_scale, causal, window_size, alibi_slopes, deterministic)
    return out if not return_softmax else (out, softmax_lse, S_dmask)


def _dense_bwd(ctx, dout, dq, dk, dv):
    q, k, v, out, softmax_lse, rng_state = ctx.saved_tensors
    _flash_attn_backward(dout, q, k, v, out, softmax_lse, dq, dk, dv, ctx.dropout_p,
                         ctx.softmax_scale, ctx.causal, ctx.window_size, ctx.alibi_slopes,
                         ctx.deterministic, rng_state=rng_state)


def _varlen_fwd(ctx, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p,
                softmax_scale, causal, window_size, alibi_slopes, deterministic, return_softmax,
                block_table, out):
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    out, q, k, v, out_padded, softmax_lse, S_dmask, rng_state = _flash_attn_varlen_forward(
        q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale,
        causal=causal, window_size=window_size, alibi_slopes=alibi_slopes,
        return_softmax=return_softmax and dropout_p > 0, block_table=block_table, out=out)
    ctx.save_for_backward(q, k, v, out_padded, softmax_lse, cu_seqlens_q, cu_seqlens_k, rng_state)
    ctx.max_seqlen_q = max_seqlen_q
    ctx.max_seqlen_k = max_seqlen_k
    _save_cfg(ctx, dropout_p, softmax_scale, causal, window_size, alibi_slopes, deterministic)
    return out if not return_softmax else (out, softmax_lse, S_dmask)


def _varlen_bwd(ctx, dout, dq, dk, dv):
    q, k, v, out, softmax_lse, cu_seqlens_q, cu_seqlens_k, rng_state = ctx.saved_tensors
    _flash_attn_varlen_backward(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q,
                                cu_seqlens_k, ctx.max_seqlen_q, ctx.max_seqlen_k, ctx.dropout_p,
                                ctx.softmax_scale, ctx.causal, ctx.window_size, ctx.alibi_slopes,
                                ctx.deterministic, rng_state=rng_state)


class FlashAttnQKVPackedFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, dropout_p, softmax_scale, causal, window_size, alibi_slopes, deterministic,
                return_softmax, out=None):
        return _dense_fwd(ctx, qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], dropout_p, softmax_scale,
                          causal, window_size, alibi_slopes, deterministic, return_softmax, out)

    @staticmethod
    def backward(ctx, dout, *args):
        q = ctx.saved_tensors[0]
        dqkv = torch.empty(q.shape[:-2] + (3, *q.shape[-2:]), dtype=q.dtype, device=q.device)
        _dense_bwd(ctx, dout, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2])
        dqkv = dqkv[..., : dout.shape[-1]]  # the head dimension may have been padded
        return (dqkv,) + (None,) * 8


class FlashAttnVarlenQKVPackedFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, cu_seqlens, max_seqlen, dropout_p, softmax_scale, causal, window_size,
                alibi_slopes, deterministic, return_softmax, out=None):
        return _varlen_fwd(ctx, qkv[:, 0], qkv[:, 1], qkv[:, 2], cu_seqlens, cu_seqlens, max_seqlen,
                           max_seqlen, dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                           deterministic, return_softmax, None, out)

    @staticmethod
    def backward(ctx, dout, *args):
        q = ctx.saved_tensors[0]
        dqkv = torch.empty(q.shape[:-2] + (3, *q.shape[-2:]), dtype=q.dtype, device=q.device)
        _varlen_bwd(ctx, dout, dqkv[:, 0], dqkv[:, 1], dqkv[:, 2])
        dqkv = dqkv[..., : dout.shape[-1]]
        return (dqkv,) + (None,) * 10


class FlashAttnKVPackedFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                deterministic, return_softmax, out=None):
        return _dense_fwd(ctx, q, kv[:, :, 0], kv[:, :, 1], dropout_p, softmax_scale, causal,
                          window_size, alibi_slopes, deterministic, return_softmax, out)

    @staticmethod
    def backward(ctx, dout, *args):
        q, k = ctx.saved_tensors[0], ctx.saved_tensors[1]
        dq = torch.empty_like(q)
        dkv = torch.empty(k.shape[:-2] + (2, *k.shape[-2:]), dtype=k.dtype, device=k.device)
        _dense_bwd(ctx, dout, dq, dkv[:, :, 0], dkv[:, :, 1])
        dq = dq[..., : dout.shape[-1]]
        dkv = dkv[..., : dout.shape[-1]]
        return (dq, dkv) + (None,) * 8


class FlashAttnVarlenKVPackedFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p,
                softmax_scale, causal, window_size, alibi_slopes, deterministic, return_softmax,
                out=None):
        return _varlen_fwd(ctx, q, kv[:, 0], kv[:, 1], cu_seqlens_q, cu_seqlens_k, max_seqlen_q,
                           max_seqlen_k, dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                           deterministic, return_softmax, None, out)

    @staticmethod
    def backward(ctx, dout, *args):
        q, k = ctx.saved_tensors[0], ctx.saved_tensors[1]
        dq = torch.empty_like(q)
        dkv = torch.empty(k.shape[:-2] + (2, *k.shape[-2:]), dtype=k.dtype, device=k.device)
        _varlen_bwd(ctx, dout, dq, dkv[:, 0], dkv[:, 1])
        dq = dq[..., : dout.shape[-1]]
        dkv = dkv[..., : dout.shape[-1]]
        return (dq, dkv) + (None,) * 12


class FlashAttnFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                deterministic, return_softmax, out=None):
        return _dense_fwd(ctx, q, k, v, dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                          deterministic, return_softmax, out)

    @staticmethod
    def backward(ctx, dout, *args):
        q, k, v = ctx.saved_tensors[:3]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        _dense_bwd(ctx, dout, dq, dk, dv)
        d = dout.shape[-1]
        return (dq[..., :d], dk[..., :d], dv[..., :d]) + (None,) * 8


class FlashAttnVarlenFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p,
                softmax_scale, causal, window_size, alibi_slopes, deterministic, return_softmax,
                block_table, out=None):
        return _varlen_fwd(ctx, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                           dropout_p, softmax_scale, causal, window_size, alibi_slopes, deterministic,
                           return_softmax, block_table, out)

    @staticmethod
    def backward(ctx, dout, *args):
        q, k, v = ctx.saved_tensors[:3]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        _varlen_bwd(ctx, dout, dq, dk, dv)
        d = dout.shape[-1]
        return (dq[..., :d], dk[..., :d], dv[..., :d]) + (None,) * 13


def flash_attn_qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False,
                              window_size=(-1, -1), alibi_slopes=None, deterministic=False,
                              return_attn_probs=False, *, out=None):
    """Attention on a packed qkv of shape (batch_size, seqlen, 3, nheads, headdim).

    dropout_p drops attention probabilities with the reference kernel's Philox pattern (state from
    torch's CUDA generator; set it to 0.0 for evaluation).  window_size=(left, right) != (-1, -1)
    selects sliding-window attention; alibi_slopes is (nheads,) or (batch_size, nheads) fp32.
    Returns out (batch_size, seqlen, nheads, headdim), plus (softmax_lse, S_dmask) when
    return_attn_probs=True (S_dmask only with dropout_p > 0, for testing: per 64-key tile
    exp(score - running max), negative where dropped).  Same contract as the reference function of this name (:658-713)."""
    return FlashAttnQKVPackedFunc.apply(qkv, dropout_p, softmax_scale, causal, window_size,
                                        alibi_slopes, deterministic, return_attn_probs, out)


def flash_attn_kvpacked_func(q, kv, dropout_p=0.0, softmax_scale=None, causal=False,
                             window_size=(-1, -1), alibi_slopes=None, deterministic=False,
                             return_attn_probs=False, *, out=None):
    """Attention with q (batch_size, seqlen_q, nheads, headdim) and packed kv
    (batch_size, seqlen_k, 2, nheads_k, headdim); nheads must be a multiple of nheads_k (MQA/GQA:
    query head i uses kv head i // (nheads // nheads_k)).  The causal mask is aligned to the
    bottom-right corner; a fully masked row yields zero output.  Reference: :716-790."""
    return FlashAttnKVPackedFunc.apply(q, kv, dropout_p, softmax_scale, causal, window_size,
                                       alibi_slopes, deterministic, return_attn_probs, out)


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                    alibi_slopes=None, deterministic=False, return_attn_probs=False, *, out=None):
    """softmax(q k^T * softmax_scale + bias) v for q (batch_size, seqlen_q, nheads, headdim) and
    k, v (batch_size, seqlen_k, nheads_k, headdim), fp16 or bf16 on an MI355X.

    softmax_scale defaults to headdim ** -0.5.  causal=True masks keys j > i + seqlen_k - seqlen_q
    (bottom-right aligned).  window_size=(l, r): query i attends keys in
    [i + seqlen_k - seqlen_q - l, i + seqlen_k - seqlen_q + r].  alibi_slopes adds
    -slope * |i + seqlen_k - seqlen_q - j|.  Returns out (batch_size, seqlen_q, nheads, headdim) and,
    with return_attn_probs=True, (out, softmax_lse (batch_size, nheads, seqlen_q), S_dmask) where S_dmask
    is an empty tensor unless dropout_p > 0.
    Reference: :793-867."""
    if _cpu_fp32 and not q.is_cuda and q.dtype == torch.float32 and torch.is_grad_enabled() \
            and any(x.requires_grad for x in (q, k, v)):
        raise RuntimeError("the fp32 CPU path of flash_attn_func is forward only: detach the inputs or call it "
                           "under torch.no_grad() (the backward kernels run on CUDA tensors)")
    return FlashAttnFunc.apply(q, k, v, dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                               deterministic, return_attn_probs, out)


def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None,
                                     causal=False, window_size=(-1, -1), alibi_slopes=None,
                                     deterministic=False, return_attn_probs=False, *, out=None):
    """Ragged-batch attention on packed qkv (total, 3, nheads, headdim); cu_seqlens (batch_size+1,)
    int32 holds the cumulative sequence lengths.  Reference: :870-932."""
    return FlashAttnVarlenQKVPackedFunc.apply(qkv, cu_seqlens, max_seqlen, dropout_p, softmax_scale,
                                              causal, window_size, alibi_slopes, deterministic,
                                              return_attn_probs, out)


def flash_attn_varlen_kvpacked_func(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                                    dropout_p=0.0, softmax_scale=None, causal=False,
                                    window_size=(-1, -1), alibi_slopes=None, deterministic=False,
                                    return_attn_probs=False, *, out=None):
    """Ragged-batch attention with q (total_q, nheads, headdim) and packed kv
    (total_k, 2, nheads_k, headdim).  Reference: :935-1023."""
    return FlashAttnVarlenKVPackedFunc.apply(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q,
                                             max_seqlen_k, dropout_p, softmax_scale, causal,
                                             window_size, alibi_slopes, deterministic,
                                             return_attn_probs, out)


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                           dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                           alibi_slopes=None, deterministic=False, return_attn_probs=False,
                           block_table=None, *, out=None):
    """Ragged-batch attention: q (total_q, nheads, headdim), k / v (total_k, nheads_k, headdim), or a
    paged KV cache (num_blocks, page_block_size, nheads_k, headdim) addressed through block_table
    (batch_size, max_blocks) int32.  cu_seqlens_q / cu_seqlens_k are (batch_size+1,) int32 cumulative
    lengths.  Returns out (total_q, nheads, headdim); softmax_lse is (batch_size, nheads,
    max_seqlen_q).  Reference: :1026-1116."""
    return FlashAttnVarlenFunc.apply(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                                     dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                                     deterministic, return_attn_probs, block_table, out)


def flash_attn_with_kvcache(q, k_cache, v_cache, k=None, v=None, rotary_cos=None, rotary_sin=None,
                            cache_seqlens: Optional[Union[int, torch.Tensor]] = None,
                            cache_batch_idx: Optional[torch.Tensor] = None,
                            block_table: Optional[torch.Tensor] = None, softmax_scale=None,
                            causal=False, window_size=(-1, -1), rotary_interleaved=True,
                            alibi_slopes=None, num_splits=0, *, out=None):
    """Attention of q (batch_size, seqlen_q, nheads, headdim) against a KV cache
    (batch_size_cache, seqlen_cache, nheads_k, headdim), or a paged cache (num_blocks,
    page_block_size, nheads_k, headdim) with block_table.

    If k / v (batch_size, seqlen_new, nheads_k, headdim) are given they are written into the cache
    IN PLACE at positions cache_seqlens .. cache_seqlens + seqlen_new, optionally rotated with
    rotary_cos / rotary_sin (seqlen_ro, rotary_dim / 2) (the query is rotated too), and attended.
    cache_seqlens is an int or (batch_size,) int32; cache_batch_idx (batch_size,) int32 selects
    cache rows.  num_splits: 0 picks the split count by heuristic, 1 disables splitting along the
    keys.  Returns out (batch_size, seqlen_q, nheads, headdim).  No backward.  Reference: :1119-1253."""
    assert k_cache.stride(-1) == 1, "k_cache must have contiguous last dimension"
    assert v_cache.stride(-1) == 1, "v_cache must have contiguous last dimension"
    q, k, v = (_last_dim_contiguous(x) for x in (q, k, v))
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    if cache_seqlens is not None and isinstance(cache_seqlens, int):
        cache_seqlens = torch.full((k_cache.shape[0],), cache_seqlens, dtype=torch.int32,
                                   device=k_cache.device)
        cache_seqlens = _last_dim_contiguous(cache_seqlens)
    cache_batch_idx = _last_dim_contiguous(cache_batch_idx)
    block_table = _last_dim_contiguous(block_table)
    out, softmax_lse = flash_attn_cuda.fwd_kvcache(
        q, k_cache, v_cache, k, v, cache_seqlens, rotary_cos, rotary_sin, cache_batch_idx,
        block_table, alibi_slopes, out, softmax_scale, causal, window_size[0], window_size[1],
        rotary_interleaved, num_splits)
    return out
