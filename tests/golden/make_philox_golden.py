"""Builds philox_aten.cpp against the installed PyTorch headers and writes philox_aten.json
(known-answer vectors for the oracle's Philox-4x32, 7 and 10 rounds)."""
import os
import subprocess
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
inc = os.path.join(os.path.dirname(torch.__file__), "include")
with tempfile.TemporaryDirectory() as td:
    exe = os.path.join(td, "philox_aten")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", inc, os.path.join(HERE, "philox_aten.cpp"), "-o", exe])
    out = subprocess.check_output([exe], text=True)
with open(os.path.join(HERE, "philox_aten.json"), "w") as f:
    f.write(out)
print("wrote philox_aten.json")
