"""Builds tests/hip/libmfma_probe.so (test-only HIP probe) with hipcc for gfx950."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "mfma_probe.hip")
LIB = os.path.join(HERE, "libmfma_probe.so")


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
                               "-fno-gpu-rdc", "-shared", "-o", LIB, SRC])
    return LIB


if __name__ == "__main__":
    print(build())
