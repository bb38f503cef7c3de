"""Builds libvqhip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m vq_vae_amd.build            # rebuild when a source is newer than the library
    python -m vq_vae_amd.build --force

hipcc cross-compiles without a GPU; the resulting .so is what the GPU box loads.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libvqhip.so")
SOURCES = ["vq_search.hip", "vq_filter.hip", "vq_ema.hip", "vq_scatter_sorted.hip", "vq_stats.hip", "vq_boundary.hip",
           "vq_conv_wgrad.hip"]
HEADERS = [os.path.join(CSRC, "vq_common.h"), os.path.join(os.path.dirname(PKG), "include", "vq_hip.h")]
# -ffp-contract=off: the kernels spell out every fused multiply-add they want; nothing else may fuse,
# or the distances would stop matching the documented rounding.
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall",
         "-Wno-unused-function", "-fno-gpu-rdc"]


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not (force or stale()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
