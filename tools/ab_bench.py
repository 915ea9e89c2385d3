#!/usr/bin/env python3
"""bench.py on the library named by VFA_LIB (A/B runs of two builds on one box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cabi
if os.environ.get("VFA_LIB"):
    cabi.LIB_PATH = os.environ["VFA_LIB"]
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
