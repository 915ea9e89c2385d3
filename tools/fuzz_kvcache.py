#!/usr/bin/env python3
"""One-off seeded sweep of flash_attn_with_kvcache against the oracle: decode and chunked shapes,
append of new k / v (in place), rotary (both layouts, partial rotary dims), paged caches or
cache_batch_idx, causal / window / ALiBi, forced split counts.  usage: fuzz_kvcache.py SEED N"""
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
bad = 0
t0 = time.time()
for it in range(n):
    dtn = str(rng.choice(["fp16", "bf16"]))
    dt = torch.float16 if dtn == "fp16" else torch.bfloat16
    b = int(rng.integers(1, 5))
    d = int(rng.choice([64, 128, 128, 128, 96, 256]))
    hk = int(rng.choice([1, 2, 4]))
    h = hk * int(rng.choice([1, 2, 4, 8]))
    sq = int(rng.choice([1, 1, 1, 2, 5, 17, 64, 130]))
    paged = bool(rng.integers(0, 2))
    page = int(rng.choice([16, 32, 64, 256]))
    sk = int(rng.choice([64, 128, 200, 256, 513, 1024, 1300]))
    if paged:
        sk = (sk + page - 1) // page * page
    snew = int(rng.choice([0, 1, sq])) if sq <= sk else 0
    snew = min(snew, sk)
    cs = torch.tensor([int(rng.integers(0, sk - snew + 1)) for _ in range(b)], dtype=torch.int32, device="cuda")
    rotary = snew > 0 and bool(rng.integers(0, 2))
    rd = int(rng.choice([d, d // 2])) // 16 * 16
    interleaved = bool(rng.integers(0, 2))
    causal = bool(rng.integers(0, 4) != 0)
    window = (-1, -1)
    if rng.integers(0, 4) == 0:
        window = (int(rng.integers(0, 300)), 0 if causal else int(rng.integers(0, 50)))
    alibi = (torch.rand(b, h, device="cuda") * 0.3) if rng.integers(0, 4) == 0 else None
    splits = int(rng.choice([0, 0, 1, 3]))
    torch.manual_seed(seed * 100003 + it)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=dt)
    knew = torch.randn(b, snew, hk, d, device="cuda", dtype=dt) if snew else None
    vnew = torch.randn(b, snew, hk, d, device="cuda", dtype=dt) if snew else None
    cos = sin = None
    if rotary:
        ang = torch.rand(sk, rd // 2, device="cuda") * 6.28
        cos, sin = torch.cos(ang).to(dt), torch.sin(ang).to(dt)
    if paged:
        per = sk // page
        nb = b * per + 3
        kc = torch.randn(nb, page, hk, d, device="cuda", dtype=dt)
        vc = torch.randn(nb, page, hk, d, device="cuda", dtype=dt)
        bt = torch.randperm(nb)[: b * per].reshape(b, per).to(torch.int32).cuda()
        cbi = None
    else:
        extra = int(rng.integers(0, 3))
        kc = torch.randn(b + extra, sk, hk, d, device="cuda", dtype=dt)
        vc = torch.randn(b + extra, sk, hk, d, device="cuda", dtype=dt)
        bt = None
        cbi = torch.randperm(b + extra)[:b].to(torch.int32).cuda() if (extra or rng.integers(0, 2)) else None
    desc = dict(dt=dtn, b=b, d=d, h=h, hk=hk, sq=sq, sk=sk, snew=snew, cs=cs.tolist(), rotary=rotary, rd=rd,
                inter=interleaved, paged=paged, page=page, causal=causal, window=window, alibi=alibi is not None,
                splits=splits, cbi=None if cbi is None else cbi.tolist())
    try:
        res = fa_oracle.attention_fwd(np_(q), np_(kc), np_(vc), dtype=dtn, causal=causal, window=window,
                                      alibi_slopes=np_(alibi), cache_seqlens=np_(cs), cache_batch_idx=np_(cbi),
                                      block_table=np_(bt), knew=np_(knew), vnew=np_(vnew), rotary_cos=np_(cos),
                                      rotary_sin=np_(sin), rotary_interleaved=interleaved)
        ro = res[0]
        out = fa.flash_attn_with_kvcache(q, kc, vc, k=knew, v=vnew, rotary_cos=cos, rotary_sin=sin, cache_seqlens=cs,
                                         cache_batch_idx=cbi, block_table=bt, causal=causal, window_size=window,
                                         rotary_interleaved=interleaved, alibi_slopes=alibi, num_splits=splits)
        o = np_(out)
        assert np.isfinite(o).all(), "non-finite output"
        err = np.abs(o - ro).max()
        assert err <= 4 * ulp_tol(dtn, ro), f"out err {err}"
        if snew:
            rk, rv = res[2], res[3]
            assert np.array_equal(np_(vc), rv), "appended V differs"
            eps = 2.0 ** -10 if dtn == "fp16" else 2.0 ** -7
            assert np.abs(np_(kc) - rk).max() <= eps * 8, "appended K differs"
    except Exception:
        bad += 1
        print("FAIL", desc, flush=True)
        traceback.print_exc(limit=1)
    if it % 50 == 0:
        print(f"{it} / {n}  {time.time() - t0:.0f} s", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
