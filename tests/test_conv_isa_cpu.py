"""Static check of the counted stage wait of vq_conv1x1_kernel<D, NCHW, FULL = true> (no GPU needed; hipcc cross-compiles).

The kernel lets a stage's barrier wait for the LDS-DMA of the next stage only, not for the stores
issued after it: `s_waitcnt vmcnt(n)` with n = the number of store instructions a wave issues behind
its DMA instructions.  Vector memory operations retire in issue order, so the wait is exact only if
the compiler emitted exactly n stores there.  This test compiles the source to gfx950 assembly and
counts them, for every instantiation of the kernel.
"""
import os
import re
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "vq_vae_amd", "csrc", "vq_search.hip")
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-unused-function", "-fno-gpu-rdc"]


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "vq_search.s"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    subprocess.check_call([hipcc, *FLAGS, "-S", "--cuda-device-only", "-o", str(out), SRC], stderr=subprocess.DEVNULL)
    return out.read_text()


def test_counted_waits_match_the_stores_behind_the_dma(assembly):
    kernels = re.findall(r"^(_ZN\S*vq_conv1x1_kernelILi\d+ELb[01]ELb1EE\S*):\s*(?:;.*)?$", assembly, re.M)
    assert len(kernels) == 10, kernels          # 5 channel counts x 2 layouts
    for name in kernels:
        start = assembly.index(name + ":")
        body = assembly[start: assembly.index(".end_amdhsa_kernel", start)]
        ops = [l.split()[0:2] for l in (x.strip() for x in body.split("\n")) if l and not l.startswith((";", "."))]
        seen = 0
        for i, op in enumerate(ops):
            if op[0] != "s_barrier":
                continue
            # the wait that guards this barrier
            j = i - 1
            while j >= 0 and ops[j][0] != "s_waitcnt":
                j -= 1
            m = re.match(r"vmcnt\((\d+)\)", ops[j][1]) if j >= 0 and len(ops[j]) > 1 else None
            if m is None or int(m.group(1)) == 0:
                continue                        # the prologue's full wait
            n = int(m.group(1))
            # walk back to the DMA block of this stage and count the stores behind it
            stores, k = 0, j
            while k >= 0 and not ops[k][0].startswith("global_load_lds"):
                if ops[k][0].startswith("global_store"):
                    stores += 1
                assert ops[k][0] != "s_barrier" or k == i, f"{name}: no DMA between two barriers"
                k -= 1
            assert k >= 0 and stores == n, f"{name}: vmcnt({n}) guards a barrier with {stores} stores behind the DMA"
            seen += 1
        assert seen >= 2, f"{name}: expected the first-stage and the steady-state counted waits, found {seen}"


def test_weights_stationary_kernel_counts_three_tiles_behind(assembly):
    """vq_conv1x1_ws_kernel waits, before the barrier that opens a tile, for all vector memory operations but
    those issued behind the tile's own DMA: the stores of three tiles and the DMA instructions of two.
    Per instantiation: the counted wait equals 3 x (stores per tile) + 2 x (DMA instructions per tile) as
    the compiler emitted them, and no other vector memory operation follows the first barrier."""
    kernels = re.findall(r"^(_ZN\S*vq_conv1x1_ws_kernel\S*):\s*(?:;.*)?$", assembly, re.M)
    assert len(kernels) == 6, kernels           # 3 channel counts x 2 layouts
    for name in kernels:
        start = assembly.index(name + ":")
        body = assembly[start: assembly.index(".end_amdhsa_kernel", start)]
        ops = [l.split() for l in (x.strip() for x in body.split("\n")) if l and not l.startswith((";", "."))]
        groups, waits, seen_barrier = [], [], False
        for n_op, op in enumerate(ops):
            kind = None
            if op[0].startswith("global_load_lds"):
                kind = "dma"
            elif op[0].startswith("global_store"):
                kind = "store"
            elif op[0].startswith(("global_", "buffer_", "flat_", "scratch_")):
                assert not seen_barrier, f"{name}: {op[0]} after the first barrier"
                kind = "other"
                waits.clear()                   # waits so far belong to the prologue (weights, bias)
            elif op[0] == "s_barrier":
                seen_barrier = True
            elif op[0] == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", " ".join(op))
                if m and int(m.group(1)) > 0:
                    waits.append(int(m.group(1)))   # counted waits behind the prologue's loads: the loop's
            if kind is None:
                if op[0].startswith(("s_cbranch", "s_branch", "s_barrier")) or op[0] == "s_waitcnt":
                    groups.append(None)         # a group ends at control flow, a wait or a barrier
                continue
            if groups and groups[-1] is not None and groups[-1][0] == kind:
                groups[-1][1] += 1
            else:
                groups.append([kind, 1])
        dma = min(g[1] for g in groups if g and g[0] == "dma")
        stores = min(g[1] for g in groups if g and g[0] == "store")
        assert waits and all(w == 3 * stores + 2 * dma for w in waits), (name, waits, stores, dma)
