"""Build libevcont_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: plain
``hipcc -shared -fPIC`` so the library is a C-ABI object usable from any host
language (``include/evcont_hip.h``)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# EVC_BUILD_TAG=<tag>: an instrumented / experimental build goes to csrc/_build_<tag>/ and libevcont_hip_<tag>.so and is
# selected with EVCONT_HIP_LIB -- the product library and its objects are never overwritten by such a build
TAG = os.environ.get("EVC_BUILD_TAG", "")
if os.environ.get("EVC_DEBUG_STAMPS") and not TAG:
    TAG = "stamps"
LIB = os.path.join(HERE, f"libevcont_hip_{TAG}.so" if TAG else "libevcont_hip.so")
OBJDIR = os.path.join(CSRC, f"_build_{TAG}") if TAG else CSRC
SOURCES = ["gemv_stream.hip", "gemv_mfma.hip", "gemv_lds.hip", "transform.hip", "pack.hip", "y2.hip", "ip1.hip", "pair_dma.hip", "pair64.hip", "dense_small.hip", "subspace_big.hip", "response.hip", "pipeline.hip"]
HEADERS = ["common.hpp", "kernels.hpp", "few_roots.hpp", os.path.join("..", "..", "include", "evcont_hip.h")]
ARCH = "gfx950"
# EVC_DEBUG_STAMPS=1 builds the eigen-kernels with their phase stamps (tools/micro/loewdin_time.py); never shipped
EXTRA = ["-DEVC_DEBUG_STAMPS"] if os.environ.get("EVC_DEBUG_STAMPS") else []
# FP64 MFMA with its accumulator in ArchVGPRs issues every 64 cycles on gfx950, with an AccVGPR accumulator (what the
# compiler's heuristic picks as soon as a kernel is register-hungry) only every ~105: 77 against 47 TFLOP/s
# (tools/micro/mfma_f64_peak.hip built both ways, profiles/mfma_f64_peak.txt).  EVC_MFMA_AGPR=1 builds the old form.
EXTRA += os.environ.get("EVC_EXTRA_DEFS", "").split()   # experiments: extra -D flags
if not os.environ.get("EVC_MFMA_AGPR"):
    EXTRA += ["-mllvm", "-amdgpu-mfma-vgpr-form"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile (if stale) and return the path of the shared library."""
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    os.makedirs(OBJDIR, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
                   "-Wall", "-Wno-unused-function"] + EXTRA + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out:
            print(out.decode(errors="replace"))
    if force or procs or _stale(LIB, objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
