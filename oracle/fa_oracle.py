"""ctypes front-end of the CPU oracle (oracle/fa_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does.  Arrays are numpy float32 (values already rounded to the 16-bit type
by the caller); `dtype` selects where the oracle rounds like the reference kernel does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libfa_oracle.so")

ROUND = {"fp32": 0, "float32": 0, "fp16": 1, "float16": 1, "bf16": 2, "bfloat16": 2}


class _Params(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p),
        ("lse", C.c_void_p), ("dout", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p),
        ("dv", C.c_void_p), ("softmax_d", C.c_void_p),
        ("b", C.c_int), ("h", C.c_int), ("h_k", C.c_int), ("d", C.c_int),
        ("seqlen_q", C.c_int), ("seqlen_k", C.c_int),
        ("cu_seqlens_q", C.c_void_p), ("cu_seqlens_k", C.c_void_p),
        ("is_seqlens_k_cumulative", C.c_int),
        ("seqused_k", C.c_void_p),
        ("block_table", C.c_void_p), ("block_table_stride", C.c_int), ("page_block_size", C.c_int),
        ("cache_batch_idx", C.c_void_p),
        ("alibi_slopes", C.c_void_p), ("alibi_batch_stride", C.c_int),
        ("softmax_scale", C.c_float),
        ("window_left", C.c_int), ("window_right", C.c_int), ("is_causal", C.c_int),
        ("round_mode", C.c_int), ("block_m", C.c_int), ("block_n", C.c_int),
        ("num_splits", C.c_int),
        ("knew", C.c_void_p), ("vnew", C.c_void_p), ("seqlen_knew", C.c_int),
        ("rotary_cos", C.c_void_p), ("rotary_sin", C.c_void_p), ("rotary_dim", C.c_int),
        ("is_rotary_interleaved", C.c_int),
        ("p_dropout", C.c_float), ("philox_seed", C.c_uint64), ("philox_offset", C.c_uint64),
        ("s_dmask", C.c_void_p), ("seqlen_q_rounded", C.c_int), ("seqlen_k_rounded", C.c_int),
        ("seqlen_ro", C.c_int),
    ]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "fa_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.fa_oracle_fwd.argtypes = [C.POINTER(_Params)]
        _lib.fa_oracle_bwd.argtypes = [C.POINTER(_Params)]
        _lib.fa_oracle_round.restype = C.c_float
        _lib.fa_oracle_round.argtypes = [C.c_float, C.c_int]
        _lib.fa_oracle_num_splits_heuristic.argtypes = [C.c_int] * 4
        _lib.fa_oracle_fold_window.argtypes = [C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4
        _lib.fa_oracle_philox.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        _lib.fa_oracle_philox.restype = None
        _lib.fa_oracle_dropout_keep.argtypes = [C.c_uint64, C.c_uint64] + [C.c_int] * 6
        _lib.fa_oracle_dropout_mask.argtypes = [C.c_uint64, C.c_uint64] + [C.c_int] * 4 + [C.c_float, C.c_void_p]
        _lib.fa_oracle_dropout_mask.restype = None
    return _lib


def _f32(x):
    return None if x is None else np.ascontiguousarray(x, dtype=np.float32)


def _i32(x):
    return None if x is None else np.ascontiguousarray(x, dtype=np.int32)


def _ptr(a):
    return None if a is None else a.ctypes.data


def fold_window(seqlen_q, seqlen_k, has_alibi, causal, window):
    c, wl, wr, loc = C.c_int(int(causal)), C.c_int(window[0]), C.c_int(window[1]), C.c_int(0)
    lib().fa_oracle_fold_window(seqlen_q, seqlen_k, int(has_alibi), C.byref(c), C.byref(wl),
                                C.byref(wr), C.byref(loc))
    return bool(c.value), wl.value, wr.value, bool(loc.value)


def num_splits_heuristic(batch_nheads_mblocks, num_sms, num_n_blocks, max_splits=128):
    return lib().fa_oracle_num_splits_heuristic(batch_nheads_mblocks, num_sms, num_n_blocks, max_splits)


def philox(seed, subsequence, offset, n_rounds=7):
    """Four 32-bit words of Philox-4x32 with counter (offset, subsequence) and key seed."""
    out = np.zeros(4, dtype=np.uint32)
    lib().fa_oracle_philox(seed, subsequence, offset, n_rounds, out.ctypes.data)
    return out


def dropout_mask(seed, offset, b, h, seqlen_q, seqlen_k, p_dropout):
    """Keep mask (b, h, seqlen_q, seqlen_k) bool of the reference's dropout for this Philox state."""
    m = np.zeros((b, h, seqlen_q, seqlen_k), dtype=np.uint8)
    lib().fa_oracle_dropout_mask(seed, offset, b, h, seqlen_q, seqlen_k, float(p_dropout), m.ctypes.data)
    return m.astype(bool)


def _round128(x):
    return (int(x) + 127) // 128 * 128


def _common(p, keep, q, k, v, dtype, softmax_scale, causal, window, alibi_slopes, cu_seqlens_q,
            cu_seqlens_k, max_seqlen_q, max_seqlen_k, seqused_k, cache_seqlens, block_table,
            cache_batch_idx):
    varlen = cu_seqlens_q is not None
    if varlen:
        b = len(cu_seqlens_q) - 1
        h, d = q.shape[1], q.shape[2]
        sq = int(max_seqlen_q)
    else:
        b, sq, h, d = q.shape
    paged = block_table is not None
    h_k = k.shape[-2]
    if paged:
        page = k.shape[1]
        bt = _i32(block_table)
        keep.append(bt)
        p.block_table, p.block_table_stride, p.page_block_size = _ptr(bt), bt.shape[1], page
        sk = int(max_seqlen_k) if varlen else bt.shape[1] * page
    elif varlen:
        sk = int(max_seqlen_k)
    else:
        sk = k.shape[1]
    p.b, p.h, p.h_k, p.d, p.seqlen_q, p.seqlen_k = b, h, h_k, d, sq, sk
    p.is_seqlens_k_cumulative = 1
    if varlen:
        cq, ck = _i32(cu_seqlens_q), _i32(cu_seqlens_k)
        keep += [cq, ck]
        p.cu_seqlens_q, p.cu_seqlens_k = _ptr(cq), _ptr(ck)
    if cache_seqlens is not None:
        cs = _i32(cache_seqlens)
        keep.append(cs)
        p.cu_seqlens_k = _ptr(cs)
        p.is_seqlens_k_cumulative = 0
    for name, arr in (("seqused_k", seqused_k), ("cache_batch_idx", cache_batch_idx)):
        if arr is not None:
            a = _i32(arr)
            keep.append(a)
            setattr(p, name, _ptr(a))
    if alibi_slopes is not None:
        al = _f32(alibi_slopes)
        keep.append(al)
        p.alibi_slopes = _ptr(al)
        p.alibi_batch_stride = al.shape[1] if al.ndim == 2 else 0
    p.softmax_scale = float(softmax_scale if softmax_scale is not None else d ** -0.5)
    p.window_left, p.window_right, p.is_causal = int(window[0]), int(window[1]), int(causal)
    p.round_mode = ROUND[dtype]
    return b, h, sq, d


def attention_fwd(q, k, v, *, dtype="fp32", softmax_scale=None, causal=False, window=(-1, -1),
                  alibi_slopes=None, cu_seqlens_q=None, cu_seqlens_k=None, max_seqlen_q=None,
                  max_seqlen_k=None, seqused_k=None, cache_seqlens=None, block_table=None,
                  cache_batch_idx=None, knew=None, vnew=None, rotary_cos=None, rotary_sin=None,
                  rotary_interleaved=True, num_splits=1, block_m=0, block_n=0, p_dropout=0.0,
                  philox_seed=0, philox_offset=0, return_softmax=False):
    """Returns (out, lse[, k_cache, v_cache][, s_dmask]).  out like q (float32 holding dtype-rounded
    values); lse (b, h, seqlen_q).  With knew/vnew the caches are updated on copies that are
    returned.  With return_softmax, s_dmask (b, h, seqlen_q_rounded, seqlen_k_rounded) is appended."""
    q, k, v = _f32(q), _f32(k), _f32(v)
    append = knew is not None
    if append:
        k, v = k.copy(), v.copy()
    p, keep = _Params(), []
    b, h, sq, d = _common(p, keep, q, k, v, dtype, softmax_scale, causal, window, alibi_slopes,
                          cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, seqused_k,
                          cache_seqlens, block_table, cache_batch_idx)
    out = np.zeros_like(q)
    lse = np.full((b, h, sq), np.inf, dtype=np.float32)
    p.q, p.k, p.v, p.out, p.lse = _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse)
    p.num_splits, p.block_m, p.block_n = int(num_splits), int(block_m), int(block_n)
    if append:
        kn, vn = _f32(knew), _f32(vnew)
        keep += [kn, vn]
        p.knew, p.vnew, p.seqlen_knew = _ptr(kn), _ptr(vn), kn.shape[1]
        if rotary_cos is not None:
            rc, rs = _f32(rotary_cos), _f32(rotary_sin)
            keep += [rc, rs]
            p.rotary_cos, p.rotary_sin = _ptr(rc), _ptr(rs)
            p.rotary_dim = rc.shape[1] * 2
            p.seqlen_ro = rc.shape[0]
            p.is_rotary_interleaved = int(rotary_interleaved)
    p.p_dropout, p.philox_seed, p.philox_offset = float(p_dropout), int(philox_seed), int(philox_offset)
    sdm = None
    if return_softmax:
        p.seqlen_q_rounded, p.seqlen_k_rounded = _round128(p.seqlen_q), _round128(p.seqlen_k)
        sdm = np.zeros((b, h, p.seqlen_q_rounded, p.seqlen_k_rounded), dtype=np.float32)
        p.s_dmask = _ptr(sdm)
    rc_ = lib().fa_oracle_fwd(C.byref(p))
    if rc_ != 0:
        raise RuntimeError(f"fa_oracle_fwd failed: {rc_}")
    res = (out, lse, k, v) if append else (out, lse)
    return res + (sdm,) if return_softmax else res


def attention_bwd(dout, q, k, v, out, lse, *, dtype="fp32", softmax_scale=None, causal=False,
                  window=(-1, -1), alibi_slopes=None, cu_seqlens_q=None, cu_seqlens_k=None,
                  max_seqlen_q=None, max_seqlen_k=None, p_dropout=0.0, philox_seed=0, philox_offset=0):
    """Returns (dq, dk, dv, softmax_d)."""
    dout, q, k, v, out, lse = (_f32(x) for x in (dout, q, k, v, out, lse))
    p, keep = _Params(), []
    b, h, sq, d = _common(p, keep, q, k, v, dtype, softmax_scale, causal, window, alibi_slopes,
                          cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, None, None, None,
                          None)
    dq, dk, dv = np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)
    sd = np.zeros((b, h, sq), dtype=np.float32)
    p.q, p.k, p.v, p.out, p.lse, p.dout = (_ptr(x) for x in (q, k, v, out, lse, dout))
    p.dq, p.dk, p.dv, p.softmax_d = _ptr(dq), _ptr(dk), _ptr(dv), _ptr(sd)
    p.p_dropout, p.philox_seed, p.philox_offset = float(p_dropout), int(philox_seed), int(philox_offset)
    rc_ = lib().fa_oracle_bwd(C.byref(p))
    if rc_ != 0:
        raise RuntimeError(f"fa_oracle_bwd failed: {rc_}")
    return dq, dk, dv, sd


def round_to(x, dtype):
    """Round a float32 array to fp16/bf16 (RNE) and return float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if ROUND[dtype] == 0:
        return x
    if ROUND[dtype] == 1:
        return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    r = np.where(nan, x.view(np.uint32), r).astype(np.uint32)
    return r.view(np.float32)
