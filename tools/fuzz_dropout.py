#!/usr/bin/env python3
"""One-off seeded sweep of the dropout variants (forward + backward through the C ABI) against the
oracle: the Philox keep mask must agree element for element, so the results do to the usual
rounding tolerances.  usage: fuzz_dropout.py SEED N"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import cabi
from oracle import fa_oracle
from gpu_util import TDT, ulp_tol
import test_gpu_random as T

seed, n = int(sys.argv[1]), int(sys.argv[2])
T.EDGE = T.EDGE + [640, 769]
rng = np.random.default_rng(seed + 1)
bad = 0
t0 = time.time()
for it, case in enumerate(T._cases(n, seed)):
    dt, b, sq, sk, h, hk, d, causal, window, alibi = case
    pd = float(rng.choice([0.1, 0.17, 0.5, 0.9]))
    ps, po = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 2 ** 20)) * 4
    try:
        torch.manual_seed(it)
        q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
        k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
        v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
        do = torch.randn_like(q)
        slopes = (torch.rand(b, h, device="cuda") * 0.3) if alibi else None
        kw = dict(causal=causal, window=window, alibi_slopes=slopes, p_dropout=pd, philox_seed=ps, philox_offset=po)
        okw = dict(dtype=dt, causal=causal, window=window, alibi_slopes=None if slopes is None else slopes.cpu().numpy(),
                   p_dropout=pd, philox_seed=ps, philox_offset=po)
        qn, kn, vn = (t.float().cpu().numpy() for t in (q, k, v))
        out, lse = cabi.fwd(q, k, v, **kw)[:2]
        ro, rl = fa_oracle.attention_fwd(qn, kn, vn, **okw)[:2]
        o = out.float().cpu().numpy()
        assert np.isfinite(o).all()
        err = np.abs(o - ro).max()
        assert err <= 2 * ulp_tol(dt, ro), f"out err {err}"
        dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, **kw)
        # differentiate both at the KERNEL's forward result: at p = 0.9 a last-place difference of O, ten
        # times amplified in delta and again in dQ / dK, would otherwise exceed the tolerance
        rdq, rdk, rdv, _ = fa_oracle.attention_bwd(do.float().cpu().numpy(), qn, kn, vn, o, lse.cpu().numpy(), **okw)
        for name, got, ref in (("dq", dq, rdq), ("dk", dk, rdk), ("dv", dv, rdv)):
            g = got.float().cpu().numpy()
            assert np.isfinite(g).all(), name
            e = np.abs(g - ref).max()
            assert e <= 4 * ulp_tol(dt, ref), f"{name} err {e} tol {4 * ulp_tol(dt, ref)}"
    except Exception:
        bad += 1
        print("FAIL", case, pd, ps, po, flush=True)
        traceback.print_exc(limit=1)
    if it % 50 == 0:
        print(f"{it} / {n}  {time.time() - t0:.0f} s", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
