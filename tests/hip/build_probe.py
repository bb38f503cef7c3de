"""Builds the test-only HIP probes under tests/hip/ (libmfma_probe.so, libvmcnt_probe.so) with hipcc for gfx950."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "mfma_probe.hip")
LIB = os.path.join(HERE, "libmfma_probe.so")
PROBES = {"mfma_probe": (SRC, LIB),
          "vmcnt_probe": (os.path.join(HERE, "vmcnt_probe.hip"), os.path.join(HERE, "libvmcnt_probe.so"))}


def build(force: bool = False) -> str:
    """Builds every probe; returns the path of the matrix-core probe (its first user's expectation)."""
    for src, lib in PROBES.values():
        if force or not os.path.exists(lib) or os.path.getmtime(src) > os.path.getmtime(lib):
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
                                   "-fno-gpu-rdc", "-shared", "-o", lib, src])
    return LIB


if __name__ == "__main__":
    print(build())
