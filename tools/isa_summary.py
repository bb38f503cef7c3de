#!/usr/bin/env python3
"""Static summary of the compiled gfx950 kernels: registers, scratch, static LDS, MFMA count per kernel,
and a check that no instruction writes through the scalar data cache (scalar stores / atomics / cache
write-back), which this code base never does.  Needs only hipcc (no GPU).

    python tools/isa_summary.py > isa_summary.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = [os.path.join(ROOT, "vq_vae_amd", "csrc", f) for f in
           ("vq_search.hip", "vq_filter.hip", "vq_ema.hip", "vq_stats.hip", "vq_boundary.hip")]
SOURCES.append(os.path.join(ROOT, "tests", "hip", "mfma_probe.hip"))
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only"]
FORBIDDEN = re.compile(r"^\s+(s_store\w*|s_buffer_store\w*|s_atomic\w*|s_buffer_atomic\w*|s_scratch_store\w*|s_dcache_wb\w*)\b", re.M)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.split("\n") if out.returncode == 0 else names


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bad = 0
    print(f"{'kernel':100s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch':>8s} {'lds':>7s} {'mfma':>5s}")
    with tempfile.TemporaryDirectory() as tmp:
        for src in SOURCES:
            asm = os.path.join(tmp, os.path.basename(src) + ".s")
            subprocess.run([hipcc, *FLAGS, "-o", asm, src], check=True, stderr=subprocess.DEVNULL)
            text = open(asm).read()
            hits = FORBIDDEN.findall(text)
            bad += len(hits)
            print(f"# {os.path.relpath(src, ROOT)}: {len(hits)} scalar-cache writes")
            bodies = dict(re.findall(r"^(_Z\w+|\w+):\s*; @\1\n(.*?)\n\s+\.end_amdhsa_kernel", text, re.S | re.M))
            rows = []
            for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
                name, d = m.group(1), m.group(2)
                get = lambda key: int(re.search(key + r" (\d+)", d).group(1))
                nv = get(r"\.amdhsa_next_free_vgpr")
                acc = re.search(r"\.amdhsa_accum_offset (\d+)", d)
                acc = int(acc.group(1)) if acc else nv
                body = bodies.get(name, "")
                rows.append((name, min(nv, acc), max(0, nv - acc), get(r"\.amdhsa_next_free_sgpr"),
                             get(r"\.amdhsa_private_segment_fixed_size"), get(r"\.amdhsa_group_segment_fixed_size"),
                             len(re.findall(r"^\s+v_mfma", body, re.M))))
            for (name, *vals), short in zip(rows, demangle([r[0] for r in rows])):
                short = re.sub(r"\(anonymous namespace\)::", "", short)
                short = re.sub(r"\(.*", "", short)[:100]
                print(f"{short:100s} {vals[0]:5d} {vals[1]:5d} {vals[2]:5d} {vals[3]:8d} {vals[4]:7d} {vals[5]:5d}")
    print(f"# total scalar-cache writes: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
