"""Build the gfx950 native libraries in-tree (no cmake, no hipify).

  liblvllm_hip.so   hand-written HIP kernels behind the C-ABI of include/lvllm_hip.h
                    (hipcc --offload-arch=gfx950, one object per .hip, linked -shared)
  _C.so             torch op registrations (`torch.ops._C`, `_C_cache_ops`,
                    `_C_cuda_utils`) forwarding to the C-ABI; host-only C++,
                    built with g++ against the torch headers.

Both land in light-vllm_amd/lib/.  Objects are cached under build/ keyed on the
source mtime, so a rebuild after touching one kernel takes seconds.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(ROOT, "build", "obj")
ARCH = "gfx950"

HIP_SOURCES = [
    "capi.hip",
    "cache.hip",
    "layernorm.hip",
    "pos_encoding.hip",
    "activation.hip",
    "attention.hip",
    "attention_bf16.hip",
    "attention_f16.hip",
    "skinny_gemm.hip",
    "prefill_attention.hip",
    "prepare_inputs.hip",
    "fp8_quant.hip",
    "sampler.hip",
]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# tuning experiments: e.g. LVLLM_EXTRA_HIPCC_FLAGS="-DLVLLM_ATTN_NBUF=2 -DLVLLM_ATTN_TUNE_ONLY"
EXTRA_FLAGS = os.environ.get("LVLLM_EXTRA_HIPCC_FLAGS", "").split()


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "lvllm_hip.h"))
    return hs


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


# Per-source flags.  The prefill kernel's softmax is vector-issue bound; without NaN honouring the
# compiler drops the canonicalising v_max x,x,x it otherwise puts in front of every max of an
# MFMA result (3 instructions per max -> 1).  Masked / out-of-range keys are removed by selects,
# not by NaN propagation, so results on finite data are unchanged.
PER_SOURCE_FLAGS = {"prefill_attention.hip": ["-fno-honor-nans"]}


def build_kernels(verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = _headers()
    jobs = []
    objs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if _newer(o, [s] + hdrs):
            jobs.append([HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
                         "-ffp-contract=off", "-Wall", "-Wno-unused-function"] +
                        PER_SOURCE_FLAGS.get(src, []) + EXTRA_FLAGS + ["-c", s, "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    lib = os.path.join(LIBDIR, "liblvllm_hip.so")
    if jobs or _newer(lib, objs):
        # no rpath to /opt/rocm on purpose: inside a torch process the HIP
        # runtime torch already loaded (same SONAME) must be the one used.
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def build_torch_bindings(verbose=False):
    import torch  # noqa: F401  (header + library locations)
    from torch.utils import cpp_extension as ce

    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(CSRC, "torch_bindings.cpp")
    out = os.path.join(LIBDIR, "_C.so")
    if not _newer(out, [src, os.path.join(ROOT, "include", "lvllm_hip.h")]):
        return out
    import sysconfig

    incs = ce.include_paths("cuda") + [sysconfig.get_paths()["include"], os.path.join(ROOT, "include")]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1",
           "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_C",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-Wno-deprecated-declarations"]
    for i in incs:
        cmd += ["-I", i]
    cmd += [src, "-o", out, "-L", torch_lib, "-L", LIBDIR,
            "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-llvllm_hip",
            "-Wl,-rpath,$ORIGIN"]
    o = _run(cmd)
    if verbose and o.strip():
        print(o)
    return out


def build_all(verbose=False):
    lib = build_kernels(verbose)
    ext = build_torch_bindings(verbose)
    return lib, ext


if __name__ == "__main__":
    print(build_all(verbose="-v" in sys.argv))
