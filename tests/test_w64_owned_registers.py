"""The forward kernels that run one wave per SIMD (64 rows per wave at d <= 128, development library;
the hand-pipelined kernel of head dims above 128, shipped) keep O and Q in accumulator registers
a[0:191] that only their asm statements touch.  The compiler knows them only as clobbers, so this test compiles the kernel
to assembly and checks that no instruction the COMPILER emitted (outside the asm blocks) reads or
writes one of them: a value parked there between two statements would silently corrupt O or Q."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vllm-flash-attention_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src,symbol,n_kernels", [("fwd_prefill_w64.hip", "_ZN3vfa22fwd_prefill_w64_kernel", 2),
                                                  ("fwd_prefill_wide.hip", "_ZN3vfa23fwd_prefill_wide_kernel", 6),
                                                  ("fwd_prefill_wide_paged.hip", "_ZN3vfa23fwd_prefill_wide_kernel", 6)])
def test_compiler_never_touches_the_owned_accumulator_registers(src, symbol, n_kernels):
    tmp = tempfile.mkdtemp()
    try:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-fvisibility=hidden", "-std=c++17",
                               "-I", os.path.join(ROOT, "include"),
                               "-I", CSRC, "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only", "-w",
                               os.path.join(CSRC, src), "-o", out])
        in_asm, bad, kernels = False, [], 0
        for line in open(out):
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            elif not in_asm:
                if re.match(r"^" + symbol + r"\S*:", line):
                    kernels += 1
                for m in re.finditer(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]", line):
                    lo = int(m.group(1) or m.group(2))
                    if lo < 192 and not line.lstrip().startswith((";", ".")):
                        bad.append(line.strip())
        assert kernels >= n_kernels, "kernel symbols not found in the assembly"
        assert not bad, "compiler-emitted instructions touch owned AGPRs:\n" + "\n".join(bad[:10])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
