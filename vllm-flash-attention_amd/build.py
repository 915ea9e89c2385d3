#!/usr/bin/env python3
"""In-tree build of the MI355X FlashAttention library.

  libvfa_hip.so                 HIP kernels + C ABI (include/vfa.h), built with hipcc for gfx950
  libvfa_hip_dev.so             development build (-DVFA_DEV): + kernels no shipped dispatch takes and the
                                kernel-variant switches; loaded by A/B tools and cross-check tests only
  vllm_flash_attn_2_cuda*.so    PyTorch extension module with the reference's pybind names
                                (fwd / varlen_fwd / fwd_kvcache / bwd / varlen_bwd); a thin shim
                                over the C ABI, built with g++ against the torch headers

The counterpart of the reference's setup.py:134-219 (CUDAExtension "vllm_flash_attn_2_cuda").
Usage: python build.py [--force] [--no-torch-ext]
       python build.py --stamp     diagnostic libvfa_hip_stamp.so (-DVFA_STAMP: per-phase cycle
                                   counters in the forward tile loop, read by tools/stamp_fwd.py)
"""
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libvfa_hip.so")
HIP_SOURCES = ["api.hip", "fwd_prefill.hip", "fwd_splitkv.hip", "bwd.hip", "fwd_prefill_drop.hip",
               "bwd_drop.hip", "fwd_prefill_split.hip", "bwd_dkdv_w64.hip", "fwd_prefill_persist.hip", "bwd_dq_persist.hip",
               "bwd_dkdv_wide.hip", "fwd_prefill_wide.hip", "bwd_dq_ilp.hip", "fwd_prefill_d64.hip", "fwd_prefill_pack.hip", "fwd_prefill_d96.hip", "bwd_d96.hip", "fwd_prefill_wide_paged.hip"]
# The development library (libvfa_hip_dev.so, -DVFA_DEV) adds the kernels no shipped dispatch takes
# and the kernel-variant switches (vfa_debug_set_tuning) that select them; the A/B tools and the
# tests that check a shipped kernel against its alternative load it.  The product never does.
DEV_SOURCES = HIP_SOURCES + ["fwd_prefill_w64.hip", "bwd_dkdv_persist.hip"]
DEV_LIB = os.path.join(HERE, "libvfa_hip_dev.so")
# Kernels built for one wave per SIMD (up to 512 registers): MFMA results that vector instructions
# consume must be allocated in the arch VGPRs, not forced into the accumulator half of the file.
EXTRA_FLAGS = {"bwd_dkdv_w64.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "bwd_dkdv_wide.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "fwd_prefill_wide.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "fwd_prefill_wide_paged.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "fwd_prefill_w64.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               # The persistent dQ kernel under LLVM's ILP-first scheduler: the default one puts every LDS
               # read directly in front of its MFMA (read, wait, MFMA, read, wait, ...); this one issues
               # the reads of the next k-steps ahead and places exponentials of the next 32 keys between
               # the dQ MFMAs.  Measured (b8 s4096 h32 d128 causal, interleaved): 1.536 -> 1.462 ms, the
               # one-block-per-workgroup kernel 1.554.  The same flag costs the persistent forward 8 %
               # and makes the dK/dV kernels spill.
               "bwd_dq_persist.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
               # the one-block-per-workgroup dQ kernels at d >= 128 (d256: 1.075 -> 0.953 ms at b4 s4096 h16 causal)
               "bwd_dq_ilp.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
               # the forward at d <= 64 without packed fp32 vector instructions (b4 s2048 h16 causal +2.7 %,
               # longer sequences +4 %)
               "fwd_prefill_d64.hip": ["-fno-slp-vectorize"]}
# the dropout translation units are one-line includes of the kernel sources
EXTRA_DEPS = {"fwd_prefill_drop.hip": ["fwd_prefill.hip"], "bwd_drop.hip": ["bwd.hip"], "bwd_d96.hip": ["bwd.hip"], "fwd_prefill_wide_paged.hip": ["fwd_prefill_wide.hip"], "bwd_dq_ilp.hip": ["bwd.hip"], "fwd_prefill_d64.hip": ["fwd_prefill.hip"], "fwd_prefill_d96.hip": ["fwd_prefill.hip"], "fwd_prefill_pack.hip": ["fwd_prefill.hip"],
              "fwd_prefill_split.hip": ["fwd_prefill.hip"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-fvisibility=hidden", "-std=c++17", "-I", INCLUDE, "-I", CSRC]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(INCLUDE, "vfa.h")]


def _compile_jobs(sources, objdir, defines, force):
    """hipcc command lines of the objects of one library that are out of date."""
    os.makedirs(objdir, exist_ok=True)
    hdrs = _headers() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".inc")]
    jobs = []
    for src in sources:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _newer(o, [s] + hdrs + [os.path.join(CSRC, e) for e in EXTRA_DEPS.get(src, [])]):
            jobs.append([HIPCC] + HIP_FLAGS + EXTRA_FLAGS.get(src, []) + list(defines) + ["-c", s, "-o", o])
    return jobs


def _run_jobs(jobs):
    if not jobs:
        return
    # the translation units that take longest first (they bound the wall time of a cold build)
    slow = ("fwd_prefill_wide", "bwd.hip", "bwd_d96", "fwd_prefill.hip", "fwd_prefill_d96", "fwd_prefill_pack", "bwd_dq_ilp", "bwd_drop")
    jobs = sorted(jobs, key=lambda c: next((i for i, k in enumerate(slow) if k in c[-3]), len(slow)))
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        for r in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs):
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError("hipcc failed: " + " ".join(r.args))


def _link(lib, sources, objdir, rebuilt, force):
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in sources]
    if force or rebuilt or _newer(lib, objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def _build_lib(lib, sources, objdir, defines, force):
    jobs = _compile_jobs(sources, objdir, defines, force)
    _run_jobs(jobs)
    return _link(lib, sources, objdir, bool(jobs), force)


def build_hip(force=False):
    return _build_lib(LIB, HIP_SOURCES, OBJ, [], force)


def build_dev(force=False):
    """libvfa_hip_dev.so: every kernel and the development switches (see DEV_SOURCES)."""
    return _build_lib(DEV_LIB, DEV_SOURCES, os.path.join(OBJ, "dev"), ["-DVFA_DEV"], force)


def build_stamp():
    """Diagnostic variant of the library with in-kernel phase stamps (never loaded by the product)."""
    out = os.path.join(HERE, "libvfa_hip_stamp.so")
    od = os.path.join(OBJ, "stamp")
    os.makedirs(od, exist_ok=True)
    objs = []
    for src in DEV_SOURCES:
        o = os.path.join(od, src.replace(".hip", ".o"))
        subprocess.check_call([HIPCC] + HIP_FLAGS + EXTRA_FLAGS.get(src, []) +
                              ["-DVFA_STAMP", "-DVFA_DEV", "-c", os.path.join(CSRC, src), "-o", o])
        objs.append(o)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_variant(name, defines, sources):
    """Development builds for A/B timing: build/libvfa_<name>.so with `defines` (-D...) applied to
    `sources`, every other object taken from the development build (never loaded by the product)."""
    build_dev()
    od = os.path.join(OBJ, "variant_" + name)
    os.makedirs(od, exist_ok=True)
    objs = []
    for src in DEV_SOURCES:
        if src in sources:
            o = os.path.join(od, src.replace(".hip", ".o"))
            subprocess.check_call([HIPCC] + HIP_FLAGS + EXTRA_FLAGS.get(src, []) + ["-DVFA_DEV"] + list(defines) +
                                  ["-c", os.path.join(CSRC, src), "-o", o])
        else:
            o = os.path.join(OBJ, "dev", src.replace(".hip", ".o"))
        objs.append(o)
    out = os.path.join(OBJ, "libvfa_%s.so" % name)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def torch_ext_path():
    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    return os.path.join(HERE, "vllm_flash_attn_2_cuda" + suffix)


def build_torch_ext(force=False):
    import torch
    from torch.utils import cpp_extension

    src = os.path.join(CSRC, "torch_binding.cpp")
    out = torch_ext_path()
    if not (force or _newer(out, [src, LIB] + _headers())):
        return out
    rocm = os.environ.get("ROCM_HOME", "/opt/rocm")
    inc = cpp_extension.include_paths()
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-DTORCH_EXTENSION_NAME=vllm_flash_attn_2_cuda",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-I", INCLUDE, "-I", os.path.join(rocm, "include"),
           "-I", sysconfig.get_paths()["include"]]
    for i in inc:
        cmd += ["-isystem", i]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd += [src, "-o", out, "-L", tlib, "-L", HERE, "-L", os.path.join(rocm, "lib"),
            "-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10_hip", "-ltorch_hip",
            "-lamdhip64", "-lvfa_hip",
            "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}", f"-Wl,-rpath,{os.path.join(rocm, 'lib')}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("building vllm_flash_attn_2_cuda failed")
    return out


def build_all(force=False, torch_ext=True):
    # the objects of both libraries in one pool of compiler jobs, then the two links
    dev_obj = os.path.join(OBJ, "dev")
    j1 = _compile_jobs(HIP_SOURCES, OBJ, [], force)
    j2 = _compile_jobs(DEV_SOURCES, dev_obj, ["-DVFA_DEV"], force)
    _run_jobs(j1 + j2)
    lib = _link(LIB, HIP_SOURCES, OBJ, bool(j1), force)
    ext = build_torch_ext(force) if torch_ext else None
    _link(DEV_LIB, DEV_SOURCES, dev_obj, bool(j2), force)
    return lib, ext


if __name__ == "__main__":
    if "--stamp" in sys.argv:
        print(build_stamp())
        sys.exit(0)
    force = "--force" in sys.argv
    print(build_all(force, "--no-torch-ext" not in sys.argv))
