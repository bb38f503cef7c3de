"""Builds libvqhip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m vq_vae_amd.build            # rebuild unless the library was built from exactly these sources
    python -m vq_vae_amd.build --force

The library is current when the SHA-256 over its sources, headers and compiler flags equals the one
stored beside it (libvqhip.so.sha256) at the last build: file times play no part, so a library that is
newer than the sources but was built from other sources is rebuilt.

hipcc cross-compiles without a GPU; the resulting .so is what the GPU box loads.
"""
from __future__ import annotations

import hashlib
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


STAMP = LIB + ".sha256"


def source_hash() -> str:
    """SHA-256 over the flags and the bytes of every source and header, in a fixed order."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def stale() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build(force: bool = False, verbose: bool = True) -> str:
    if not (force or stale()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    digest = source_hash()
    if os.path.exists(STAMP):
        os.remove(STAMP)          # a failed build must not leave a stamp that vouches for an old library
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
    with open(STAMP, "w") as f:
        f.write(digest + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
