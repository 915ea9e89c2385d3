#!/usr/bin/env python3
"""One-off extended sweep of the seeded random parity tests (tests/test_gpu_random.py) with other
seeds and longer sequences: usage fuzz_gpu.py SEED N_DENSE N_VARLEN.  Prints every failing case."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_random as T

seed, n_dense, n_varlen = (int(a) for a in sys.argv[1:4])
T.EDGE = T.EDGE + [640, 769, 1024, 1100, 1537]
bad = 0
t0 = time.time()
for i, case in enumerate(T._cases(n_dense, seed)):
    try:
        T._run(case, True)
    except Exception:
        bad += 1
        print("FAIL dense", case, flush=True)
        traceback.print_exc(limit=1)
    if i % 25 == 0:
        print(f"dense {i} / {n_dense}  {time.time() - t0:.0f} s", flush=True)
for i, case in enumerate(T._varlen_cases(n_varlen, seed + 1)):
    try:
        T.test_random_varlen_fwd_bwd(case)
    except Exception:
        bad += 1
        print("FAIL varlen", case, flush=True)
        traceback.print_exc(limit=1)
    if i % 25 == 0:
        print(f"varlen {i} / {n_varlen}  {time.time() - t0:.0f} s", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
