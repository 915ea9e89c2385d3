"""ctypes binding of the C ABI (include/vfa.h) used by the GPU parity tests, bench.py and smoke()."""
import contextlib
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "vllm-flash-attention_amd", "libvfa_hip.so")
# development build: every kernel variant + vfa_debug_set_tuning (never loaded by the product)
DEV_LIB_PATH = os.path.join(ROOT, "vllm-flash-attention_amd", "libvfa_hip_dev.so")

_i64, _int, _f, _p = C.c_int64, C.c_int, C.c_float, C.c_void_p


class VfaParams(C.Structure):
    _fields_ = [
        ("dtype", _int),
        ("q", _p), ("k", _p), ("v", _p), ("out", _p), ("softmax_lse", _p),
        ("q_batch_stride", _i64), ("q_row_stride", _i64), ("q_head_stride", _i64),
        ("k_batch_stride", _i64), ("k_row_stride", _i64), ("k_head_stride", _i64),
        ("v_batch_stride", _i64), ("v_row_stride", _i64), ("v_head_stride", _i64),
        ("o_batch_stride", _i64), ("o_row_stride", _i64), ("o_head_stride", _i64),
        ("b", _int), ("seqlen_q", _int), ("seqlen_k", _int),
        ("h", _int), ("h_k", _int), ("d", _int),
        ("cu_seqlens_q", _p), ("cu_seqlens_k", _p), ("seqused_k", _p),
        ("is_seqlens_k_cumulative", _int),
        ("block_table", _p), ("block_table_batch_stride", _i64), ("page_block_size", _int),
        ("cache_batch_idx", _p),
        ("alibi_slopes", _p), ("alibi_slopes_batch_stride", _i64),
        ("p_dropout", _f), ("softmax_scale", _f),
        ("is_causal", _int), ("window_size_left", _int), ("window_size_right", _int),
        ("num_splits", _int), ("softmax_lseaccum", _p), ("oaccum", _p),
        ("knew", _p), ("vnew", _p),
        ("knew_batch_stride", _i64), ("knew_row_stride", _i64), ("knew_head_stride", _i64),
        ("vnew_batch_stride", _i64), ("vnew_row_stride", _i64), ("vnew_head_stride", _i64),
        ("seqlen_knew", _int),
        ("rotary_cos", _p), ("rotary_sin", _p), ("rotary_dim", _int), ("is_rotary_interleaved", _int),
        ("dout", _p), ("dq", _p), ("dk", _p), ("dv", _p),
        ("do_batch_stride", _i64), ("do_row_stride", _i64), ("do_head_stride", _i64),
        ("dq_batch_stride", _i64), ("dq_row_stride", _i64), ("dq_head_stride", _i64),
        ("dk_batch_stride", _i64), ("dk_row_stride", _i64), ("dk_head_stride", _i64),
        ("dv_batch_stride", _i64), ("dv_row_stride", _i64), ("dv_head_stride", _i64),
        ("dsoftmax_sum", _p), ("dq_accum", _p), ("dk_accum", _p), ("dv_accum", _p),
        ("deterministic", _int),
        ("stream", _p),
        ("stage_events", _p),
        ("philox_seed", C.c_uint64), ("philox_offset", C.c_uint64), ("rng_state", _p),
        ("philox_seed_ptr", _p), ("philox_offset_ptr", _p), ("philox_offset_intragraph", C.c_uint32),
        ("s_dmask", _p), ("seqlen_q_rounded", _int), ("seqlen_k_rounded", _int),
        ("bwd_head_splits", _int), ("bwd_part_rows", _i64), ("seqlen_ro", _int), ("total_q", _int),
    ]


ENTRY_POINTS = ["vfa_mha_fwd", "vfa_mha_varlen_fwd", "vfa_mha_fwd_kvcache", "vfa_mha_bwd",
                "vfa_mha_varlen_bwd", "vfa_mha_fwd_kvcache_noalloc"]
NOALLOC_SCRATCH_BYTES = 32 << 20   # VFA_NOALLOC_SCRATCH_BYTES of include/vfa.h
_lib = None


def load(path):
    """Loads one build of the library and declares its entry points."""
    import torch  # noqa: F401  (first, so that both use the HIP runtime torch ships)
    h = C.CDLL(path)
    for name in ENTRY_POINTS:
        fn = getattr(h, name)
        fn.argtypes = [C.POINTER(VfaParams)]
        fn.restype = _int
    h.vfa_last_error.restype = C.c_char_p
    h.vfa_version.restype = C.c_char_p
    h.vfa_params_sizeof.restype = C.c_size_t
    h.vfa_num_splits_heuristic.argtypes = [_int] * 4
    h.vfa_splitkv_num_splits.argtypes = [_int] * 6
    h.vfa_decode_num_splits.argtypes = [_int] * 7
    h.vfa_bwd_dq_accum_elems.argtypes = [_int] * 5
    h.vfa_bwd_dq_accum_elems.restype = C.c_size_t
    h.vfa_bwd_head_splits.argtypes = [_int] * 5
    h.vfa_dispatch_geometry.argtypes = [C.POINTER(VfaParams), _int]
    h.vfa_dispatch_geometry.restype = _int
    assert h.vfa_params_sizeof() == C.sizeof(VfaParams), "ctypes mirror of vfa_params is stale"
    return h


def lib():
    """libvfa_hip.so, loaded once."""
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
    return _lib


_dev = None


def dev_lib():
    """libvfa_hip_dev.so, loaded once."""
    global _dev
    if _dev is None:
        _dev = load(DEV_LIB_PATH)
        _dev.vfa_debug_set_tuning.argtypes = [_int, _int]
        _dev.vfa_debug_set_tuning.restype = None
    return _dev


@contextlib.contextmanager
def dev(switch=None, value=0):
    """Inside the block every call of this module goes to the development library, with
    development switch `switch` set to `value` (the library's other switches stay 0)."""
    global _lib
    d = dev_lib()
    prev = lib()
    _lib = d
    if switch is not None:
        d.vfa_debug_set_tuning(switch, value)
    try:
        yield d
    finally:
        if switch is not None:
            d.vfa_debug_set_tuning(switch, 0)
        _lib = prev


class VfaError(RuntimeError):
    pass


def call(name, params):
    rc = getattr(lib(), name)(C.byref(params))
    if rc != 0:
        raise VfaError(f"{name} -> {rc}: {lib().vfa_last_error().decode()}")


def _dt(t):
    import torch
    return {torch.float16: 0, torch.bfloat16: 1}[t.dtype]


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def _set4(P, name, t, batched=True):
    setattr(P, name + "_row_stride", t.stride(-3))
    setattr(P, name + "_head_stride", t.stride(-2))
    if batched:
        setattr(P, name + "_batch_stride", t.stride(0))


def fwd(q, k, v, *, softmax_scale=None, causal=False, window=(-1, -1), alibi_slopes=None,
        cu_seqlens_q=None, cu_seqlens_k=None, max_seqlen_q=None, max_seqlen_k=None, seqused_k=None,
        cache_seqlens=None, cache_batch_idx=None, block_table=None, knew=None, vnew=None,
        rotary_cos=None, rotary_sin=None, rotary_interleaved=True, num_splits=1, kvcache=False,
        out=None, p_dropout=0.0, philox_seed=0, philox_offset=0, return_softmax=False, rng_state=None,
        noalloc=False):
    """Forward through the C ABI on torch CUDA tensors.  Returns (out, lse), plus s_dmask
    (b, h, seqlen_q_rounded, seqlen_k_rounded) with return_softmax.  Picks the entry point like the
    reference's Python layer does: varlen if cu_seqlens_q, kvcache if asked, else fwd.  noalloc:
    vfa_mha_fwd_kvcache_noalloc (split partials in the library's static device array; no split
    scratch is allocated here)."""
    import torch
    P = VfaParams()
    keep = []
    varlen = cu_seqlens_q is not None
    P.dtype = _dt(q)
    d = q.shape[-1]
    if varlen:
        b = cu_seqlens_q.numel() - 1
        h, sq = q.shape[1], int(max_seqlen_q)
    else:
        b, sq, h = q.shape[0], q.shape[1], q.shape[2]
    paged = block_table is not None
    hk = k.shape[-2]
    if paged:
        page = k.shape[1]
        sk = int(max_seqlen_k) if varlen else block_table.shape[1] * page
        P.block_table = block_table.data_ptr()
        P.block_table_batch_stride = block_table.stride(0)
        P.page_block_size = page
    elif varlen:
        sk = int(max_seqlen_k)
    else:
        sk = k.shape[1]
    if out is None:
        out = torch.empty_like(q)
    lse = torch.empty(b, h, sq, dtype=torch.float32, device=q.device)
    P.q, P.k, P.v, P.out, P.softmax_lse = (t.data_ptr() for t in (q, k, v, out, lse))
    _set4(P, "q", q, not varlen)
    _set4(P, "o", out, not varlen)
    _set4(P, "k", k, paged or not varlen)
    _set4(P, "v", v, paged or not varlen)
    P.b, P.seqlen_q, P.seqlen_k, P.h, P.h_k, P.d = b, sq, sk, h, hk, d
    P.is_seqlens_k_cumulative = 1
    if varlen:
        P.cu_seqlens_q, P.cu_seqlens_k = cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr()
        P.total_q = q.shape[0]
    if cache_seqlens is not None:
        P.cu_seqlens_k = cache_seqlens.data_ptr()
        P.is_seqlens_k_cumulative = 0
    if seqused_k is not None:
        P.seqused_k = seqused_k.data_ptr()
    if cache_batch_idx is not None:
        P.cache_batch_idx = cache_batch_idx.data_ptr()
    if alibi_slopes is not None:
        P.alibi_slopes = alibi_slopes.data_ptr()
        P.alibi_slopes_batch_stride = alibi_slopes.stride(0) if alibi_slopes.dim() == 2 else 0
    P.softmax_scale = float(softmax_scale if softmax_scale is not None else d ** -0.5)
    P.is_causal, P.window_size_left, P.window_size_right = int(causal), int(window[0]), int(window[1])
    P.num_splits = int(num_splits)
    if num_splits > 1 and not noalloc:
        lacc = torch.empty(num_splits, b, h, sq, dtype=torch.float32, device=q.device)
        oacc = torch.empty(num_splits, b, h, sq, d, dtype=torch.float32, device=q.device)
        keep += [lacc, oacc]
        P.softmax_lseaccum, P.oaccum = lacc.data_ptr(), oacc.data_ptr()
    if knew is not None:
        P.knew, P.vnew, P.seqlen_knew = knew.data_ptr(), vnew.data_ptr(), knew.shape[1]
        _set4(P, "knew", knew)
        _set4(P, "vnew", vnew)
        if rotary_cos is not None:
            P.rotary_cos, P.rotary_sin = rotary_cos.data_ptr(), rotary_sin.data_ptr()
            P.rotary_dim = rotary_cos.shape[1] * 2
            P.seqlen_ro = rotary_cos.shape[0]
            P.is_rotary_interleaved = int(rotary_interleaved)
    P.stream = _stream()
    P.p_dropout, P.philox_seed, P.philox_offset = float(p_dropout), int(philox_seed), int(philox_offset)
    if rng_state is not None:
        P.rng_state = rng_state.data_ptr()
    sdm = None
    if return_softmax:
        r128 = lambda x: (x + 127) // 128 * 128
        sdm = torch.zeros(b, h, r128(sq), r128(sk), dtype=q.dtype, device=q.device)
        P.s_dmask, P.seqlen_q_rounded, P.seqlen_k_rounded = sdm.data_ptr(), r128(sq), r128(sk)
    if noalloc:
        assert not varlen and not return_softmax
        call("vfa_mha_fwd_kvcache_noalloc", P)
    elif varlen:
        call("vfa_mha_varlen_fwd", P)
    elif kvcache or cache_seqlens is not None or paged or knew is not None or cache_batch_idx is not None:
        call("vfa_mha_fwd_kvcache", P)
    else:
        call("vfa_mha_fwd", P)
    return (out, lse, sdm) if return_softmax else (out, lse)


def bwd(dout, q, k, v, out, lse, *, softmax_scale=None, causal=False, window=(-1, -1),
        alibi_slopes=None, cu_seqlens_q=None, cu_seqlens_k=None, max_seqlen_q=None,
        max_seqlen_k=None, deterministic=False, p_dropout=0.0, philox_seed=0, philox_offset=0,
        rng_state=None, head_splits=None):
    """Backward through the C ABI.  Returns (dq, dk, dv, softmax_d)."""
    import torch
    P = VfaParams()
    varlen = cu_seqlens_q is not None
    P.dtype = _dt(q)
    d = q.shape[-1]
    if varlen:
        b = cu_seqlens_q.numel() - 1
        h, sq, sk = q.shape[1], int(max_seqlen_q), int(max_seqlen_k)
    else:
        b, sq, h, sk = q.shape[0], q.shape[1], q.shape[2], k.shape[1]
    hk = k.shape[-2]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    sq_r = (sq + 127) // 128 * 128
    sd = torch.empty(b, h, sq_r, dtype=torch.float32, device=q.device)
    P.q, P.k, P.v, P.out, P.softmax_lse = (t.data_ptr() for t in (q, k, v, out, lse))
    P.dout, P.dq, P.dk, P.dv = (t.data_ptr() for t in (dout, dq, dk, dv))
    for name, t in (("q", q), ("k", k), ("v", v), ("o", out), ("do", dout), ("dq", dq), ("dk", dk), ("dv", dv)):
        _set4(P, name, t, not varlen)
    P.b, P.seqlen_q, P.seqlen_k, P.h, P.h_k, P.d = b, sq, sk, h, hk, d
    P.is_seqlens_k_cumulative = 1
    if varlen:
        P.cu_seqlens_q, P.cu_seqlens_k = cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr()
    if alibi_slopes is not None:
        P.alibi_slopes = alibi_slopes.data_ptr()
        P.alibi_slopes_batch_stride = alibi_slopes.stride(0) if alibi_slopes.dim() == 2 else 0
    P.softmax_scale = float(softmax_scale if softmax_scale is not None else d ** -0.5)
    P.is_causal, P.window_size_left, P.window_size_right = int(causal), int(window[0]), int(window[1])
    P.dsoftmax_sum = sd.data_ptr()
    P.deterministic = int(deterministic)
    P.stream = _stream()
    P.p_dropout, P.philox_seed, P.philox_offset = float(p_dropout), int(philox_seed), int(philox_offset)
    if rng_state is not None:
        P.rng_state = rng_state.data_ptr()
    if head_splits is None:  # what the torch extension does: ask the library
        cus = torch.cuda.get_device_properties(q.device).multi_processor_count
        head_splits = lib().vfa_bwd_head_splits(b, h, hk, sk, cus)
    if head_splits > 1:
        rows = k.shape[0] if varlen else b * sk
        dka = torch.empty(head_splits, rows, hk, d, dtype=torch.float32, device=q.device)
        dva = torch.empty_like(dka)
        P.bwd_head_splits, P.bwd_part_rows = head_splits, rows
        P.dk_accum, P.dv_accum = dka.data_ptr(), dva.data_ptr()
    call("vfa_mha_varlen_bwd" if varlen else "vfa_mha_bwd", P)
    return dq, dk, dv, sd
