#!/usr/bin/env python3
"""Benchmark of the quantizer hot path: fused quantize + EMA update, K=8192, D=256, fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows ROWS_PER_GPU]

One step = one training forward of the quantizer over one batch of synthetic latents already
resident in HBM: codebook prep, MFMA nearest-code search fused with gather / straight-through /
loss, code histogram, EMA embedding sums, scalars, EMA cluster + codebook update (and, for
N > 1, the RCCL all-reduce of the EMA statistics).  Workload = benchmark configuration 2: 4 Mi latent
vectors per GPU presented as (4096, 256, 32, 32) NCHW.  N > 1 is launched by
torch.distributed.run, one rank per GPU, rows sharded data-parallel (weak scaling).

Prints ONE JSON line on rank 0 with the throughput, the roofline of the dominant kernel
(HIP events around every search launch inside the timed region) and, at N = 1, a CPU
baseline: the dense one-hot PyTorch formulation timed on the host cores on a bounded sample,
and a parity gate (one training step on a 16 Ki-row sample against the C restatement).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K, D = 8192, 256
H = W = 32
BETA, DECAY, EPS = 3.0, 0.95, 1e-5
PEAK_FP32_MFMA_TFLOPS = 157.3   # dense fp32 matrix peak of MI355X (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz)


def host_cpu() -> dict:
    """Model name, logical and physical core counts of the host, from /proc/cpuinfo."""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                key, _, val = ln.partition(":")
                key, val = key.strip(), val.strip()
                if key == "model name":
                    model = val
                elif key == "processor":
                    logical += 1
                elif key == "physical id":
                    phys = val
                elif key == "core id":
                    core = val
                elif not key and phys is not None:
                    cores.add((phys, core)); phys = core = None
    except OSError:
        pass
    return {"model": model, "logical_cpus": logical or (os.cpu_count() or 1), "physical_cores": len(cores) or None}


def cpu_baseline(z: torch.Tensor, E: torch.Tensor, reps: int = 10):
    """The reference's dense formulation (oracle/dense_port.py) on the host cores.
    Sample: the first rows of the benchmark batch (`z`, a host copy: the N x K fp32 matrices must fit host
    memory) against the benchmark's initial codebook `E`, `reps` timed steps after one warm-up.  The cost of
    the formulation is linear in the number of rows (three N x K x D products and N x K elementwise
    passes), so vectors/s at the sample size is the rate the full 4 Mi-row batch would run at, chunk after
    chunk.  Returns (the baseline entry, the indices the warm-up step chose from the fresh state): the
    latter feed the near-tie census of the parity gate."""
    from oracle.dense_port import dense_forward, new_state
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    rows = z.shape[0] * H * W
    st = new_state(E)
    dense_idx = dense_forward(z, st, BETA, DECAY, EPS, 1.0, True)[3].reshape(-1).clone()  # warm-up
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        dense_forward(z, st, BETA, DECAY, EPS, 1.0, True)
        times.append(time.perf_counter() - t0)
    med, best = float(np.median(times)), float(np.min(times))
    cpu = host_cpu()
    return {"value": rows / med, "unit": "vectors/s", "cores": threads, "kind": "port",
            "value_best": rows / best, "cpu_model": cpu["model"], "host_logical_cpus": cpu["logical_cpus"],
            "host_physical_cores": cpu["physical_cores"], "reps": reps,
            "seconds_per_step_median": med, "seconds_per_step_min": best,
            "sample": f"dense one-hot PyTorch formulation of the reference forward+EMA, K={K} D={D}, "
                      f"{rows} vectors/step, {reps} steps after 1 warm-up, value = rows / median step time "
                      f"({med:.3f} s; min {best:.3f} s), {threads} torch threads, torch {torch.__version__} CPU; "
                      f"cost is linear in the rows, so the rate holds for the 4 Mi-row batch processed in chunks "
                      f"of this size ({4 * 1024 * 1024 / rows * med:.0f} s extrapolated)"}, dense_idx


def parity_gate(zs: torch.Tensor, E0: torch.Tensor, dense_idx=None) -> dict:
    """Correctness gate printed beside the throughput: one training step from a fresh EMA state
    on the first latent vectors of the benchmark batch (`zs`), on the GPU and through the
    index-based C restatement (oracle/vq_oracle.c) on the host cores, which is timed as well.
    `dense_idx`: the indices the dense one-hot formulation (the reference's own arithmetic: sgemm +
    torch.argmin) chose for the same rows on the host; every row where the GPU disagrees with them is
    classified in float64 as a near-tie or not (tools/near_tie_census.py)."""
    from oracle import vq_oracle as vo
    from tools.near_tie_census import census
    from vq_vae_amd import ops
    dev = zs.device
    E = E0.clone()
    ema_w = torch.zeros(K, D, device=dev); cs = torch.zeros(K, device=dev); usage = torch.zeros(K, device=dev)
    out = ops.quantize_step(zs, E, ema_w, cs, usage, beta=BETA, decay=DECAY, epsilon=EPS, temperature=1.0,
                            training=True, ws=ops.Workspace(dev))
    torch.cuda.synchronize()
    st = vo.OracleState.fresh(E0.cpu().numpy())
    t0 = time.perf_counter()
    ref = vo.forward(zs.cpu().numpy(), st, BETA, DECAY, EPS, 1.0, True)
    t_cpu = time.perf_counter() - t0

    def rel(a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))

    idx = out.indices.reshape(-1).cpu().numpy()
    counts = (out.counts.cpu().numpy().astype(np.int64) & 0xFFFFFFFF)
    vs_dense = None
    if dense_idx is not None:
        vs_dense = census(zs.cpu(), E0.cpu(), out.indices.cpu(), dense_idx)
        vs_dense["against"] = "dense one-hot torch formulation on the host (MKL sgemm + torch.argmin), same rows"
    return {"rows": int(idx.size), "vs_dense_formulation": vs_dense,
            "index_mismatches": int((idx != ref["indices"].reshape(-1)).sum()),
            "histogram_equal": bool(np.array_equal(counts, np.asarray(ref["counts"]).astype(np.int64))),
            "max_rel_err": {"quantized": rel(out.quantized.cpu().numpy(), ref["quantized"]),
                            "loss": rel(out.loss.item(), ref["loss"]),
                            "perplexity": rel(out.perplexity.item(), ref["perplexity"]),
                            "codebook": rel(E.cpu().numpy(), st.embedding), "ema_w": rel(ema_w.cpu().numpy(), st.ema_w),
                            "cluster_size": rel(cs.cpu().numpy(), st.ema_cluster_size)},
            "checker": f"index-based C restatement of the reference forward, OpenMP on the host cores: "
                       f"{idx.size / t_cpu:.0f} vectors/s ({t_cpu:.2f} s for the sample)"}


def measured_traffic(rows: int, kernel_prefix: str):
    """HBM bytes per launch of the search kernel from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction).  bench.py cannot
    collect PMC counters itself, so the figure is only quoted when the profile was taken on this workload
    AND on this build of the kernel (SHA-256 of vq_search.hip recorded in the profile); None otherwise."""
    import glob
    import hashlib
    try:
        with open(os.path.join(ROOT, "vq_vae_amd", "csrc", "vq_search.hip"), "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                prof = json.load(f)
            w = prof["workload"]
            if (w["K"], w["D"], w["rows"]) != (K, D, rows):
                continue
            if prof.get("source_sha256", {}).get("vq_vae_amd/csrc/vq_search.hip") != sha:
                continue
            for k in prof["kernels"]:
                if k["kernel"].startswith(kernel_prefix):
                    return float(k["hbm_bytes_corrected"]), os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def launch_ranks(n: int) -> int:
    """Start `n` ranks of this script under torch.distributed.run on this node and wait for them."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=4 * 1024 * 1024, help="latent vectors per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-filtered", action="store_true", help="skip the extra timing of the filtered search")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only "
                                                      "to rehearse the multi-rank path on a box with fewer GPUs)")
    ap.add_argument("--codebook", choices=["normal", "ref-init"], default="normal",
                    help="normal: N(0,1) rows (high utilisation, the headline case); ref-init: the reference's "
                         "U(-1/K, 1/K) initialisation, which collapses onto a few codes (skewed EMA scatter)")
    args = ap.parse_args()

    if os.environ.get("BENCH_WATCHDOG"):
        # debugging aid: dump every thread's stack and exit if the run takes longer than this many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_WATCHDOG"]), exit=True)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Called directly with N > 1: this process has not touched the GPU yet, so it starts the N ranks as
        # fresh children (one per GPU), relays their output (rank 0 prints the JSON line) and exits with
        # their status.
        raise SystemExit(launch_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if args.backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())     # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    from vq_vae_amd import _lib, ops
    from vq_vae_amd.distributed import make_stats_reduce

    stats_reduce = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        stats_reduce = make_stats_reduce()

    rows = args.rows // (H * W) * (H * W)
    B = rows // (H * W)
    # synthetic latents N(0,1) (seed 0 + rank); codebook from a separate N(0,1) stream (seed 1, same on
    # every rank) so that codes and latents share a scale and utilisation is high
    gz = torch.Generator(device=dev).manual_seed(rank)
    z = torch.randn(B, D, H, W, device=dev, generator=gz)
    ge = torch.Generator(device=dev).manual_seed(1)
    E = torch.randn(K, D, device=dev, generator=ge)
    if args.codebook == "ref-init":
        E = (torch.rand(K, D, device=dev, generator=ge) * 2 - 1) / K       # vqvae/ema_vector_quantizer.py:22
    E0 = E.clone()
    ema_w = torch.zeros(K, D, device=dev)
    cs = torch.zeros(K, device=dev)
    usage = torch.zeros(K, device=dev)
    ws = ops.Workspace(dev)

    ev_pairs = []

    def step(timed: bool, filtered: bool = False, local: bool = False):
        ev = None
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev_pairs.append(ev)
        return ops.quantize_step(z, E, ema_w, cs, usage, beta=BETA, decay=DECAY, epsilon=EPS, temperature=1.0,
                                 training=True, ws=ws, stats_reduce=None if local else stats_reduce,
                                 search_events=ev, filtered=filtered)

    def timed_run(filtered: bool, across_ranks: bool = True) -> float:
        """W untimed + K timed steps, barrier + synchronize on both sides, max over ranks (seconds).
        across_ranks=False times this rank alone: no barrier, no max, and the EMA statistics are not
        summed over the ranks either, so the run contains no collective at all."""
        sync = world > 1 and across_ranks
        local = not across_ranks
        for _ in range(args.warmup):
            step(False, filtered, local)
        torch.cuda.synchronize()
        if sync:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True, filtered, local)
        torch.cuda.synchronize()
        if sync:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if sync:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    elapsed = timed_run(False)
    out = step(False)
    torch.cuda.synchronize()

    search_all = [a.elapsed_time(b) for a, b in ev_pairs]
    search_ms = float(np.mean(search_all))
    flops = 2.0 * rows * K * D                       # algorithmic: one N x K x D contraction per step
    achieved = flops / (search_ms * 1e-3) / 1e12
    # sanity of the last step's outputs (cheap; outside the timed region)
    assert out.indices.shape[0] == rows and bool(torch.isfinite(out.loss)) and bool(torch.isfinite(out.perplexity))

    # the same step with the filtered search (bf16 matrix-core filter + exact fp32 verification of the
    # surviving candidates): identical outputs, reported beside the fp32 headline, never as `value`
    def filtered_extra() -> dict:
        ev_pairs.clear()
        E.copy_(E0); ema_w.zero_(); cs.zero_(); usage.zero_()
        ref = ops.search(z, E, ws, want_quantized=False)["indices"].clone()
        got = ops.search(z, E, ws, want_quantized=False, filtered=True)["indices"]
        same = bool(torch.equal(ref, got))
        del ref, got
        f_elapsed = timed_run(True, across_ranks=False)   # rank-local clock: nothing here may deadlock the job
        f_search_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs]))
        return {"value": rows * world * args.steps / f_elapsed, "unit": "vectors/s",
                "ms_per_step": f_elapsed / args.steps * 1e3, "search_ms": f_search_ms,
                "indices_identical_to_fp32_search": same, "rows_compared": rows,
                "effective_tflops": 2.0 * rows * K * D / (f_search_ms * 1e-3) / 1e12,
                "timing": "max over ranks" if world == 1 else "rank 0 only, rank-local EMA statistics",
                "note": "opt-in search_mode='filtered': bf16 MFMA pass under a proven error bound selects the candidates, "
                        "fp32 chain decides; outputs bit-identical to the fp32 search"}

    filt = None
    if not args.no_filtered:
        try:
            filt = filtered_extra()
        except Exception as exc:  # the extra must never cost the headline line
            filt = {"error": f"{type(exc).__name__}: {exc}"}

    # the same step at the size of one GPU's share of a training batch (64 images -> 64 Ki latents): not
    # the headline workload, reported for orientation only
    def small_batch_extra(rows_small: int = 65536) -> dict:
        zs = z[: rows_small // (H * W)].contiguous()
        res = {"rows": rows_small}
        for name, filt in (("fp32_ms_per_step", False), ("filtered_ms_per_step", True)):
            Es = E0.clone(); ws_s = ops.Workspace(dev)
            w_s = torch.zeros(K, D, device=dev); c_s = torch.zeros(K, device=dev); u_s = torch.zeros(K, device=dev)

            def one():
                ops.quantize_step(zs, Es, w_s, c_s, u_s, beta=BETA, decay=DECAY, epsilon=EPS, temperature=1.0,
                                  training=True, ws=ws_s, filtered=filt)
            for _ in range(3):
                one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                one()
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / 20 * 1e3
        return res

    small = None
    if world == 1 and not args.no_filtered and rows >= 65536:
        try:
            small = small_batch_extra()
        except Exception as exc:
            small = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        # which of the library's two matrix-core kernels the search ran for this batch (same results either way)
        persistent = bool(_lib.load().vq_search_uses_persistent_kernel(rows, K, D, 1, H * W))
        kernel_name = "vq_search_cyclic_kernel" if persistent else "vq_search_mfma_kernel"
        traffic, traffic_src = measured_traffic(rows, kernel_name)
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "quantized latent vectors/sec (argmin+EMA update), K=8192 D=256",
            "value": rows * world * args.steps / elapsed,
            "unit": "vectors/s",
            "n_gpus": world,
            "world_size": dist.get_world_size() if world > 1 else 1,
            "backend": (dist.get_backend() if world > 1 else None),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"fused quantize+EMA update, K={K}, D={D}, {rows} latent vectors per GPU as "
                                   f"({B},{D},{H},{W}) NCHW fp32, N(0,1) latents, "
                                   + ("N(0,1) codebook" if args.codebook == "normal" else "U(-1/K,1/K) codebook (skewed)")
                                   + f", decay {DECAY}",
                       "rows_per_gpu": rows, "parallelism": f"dp{world}" if world > 1 else "single"},
            "roofline": {"bound": "mfma", "kernel": f"{kernel_name}<{D},NCHW>", "achieved": achieved,
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": traffic, "traffic_unit": "bytes/launch (rocprofv3 PMC)", "traffic_source": traffic_src,
                         "algorithmic_hbm_bytes": rows * (2 * D * 4 + 12), "ms_per_launch": search_ms,
                         "ms_per_launch_median": float(np.median(search_all)), "ms_per_launch_min": float(np.min(search_all)),
                         "flops_per_launch": flops},
            "perplexity": float(out.perplexity.item()),
        }
        if filt is not None:
            line["filtered_search"] = filt
        if small is not None:
            line["per_gpu_training_batch"] = small
        if world == 1 and not args.no_cpu_baseline:
            zs = z[: max(1, min(16384, rows) // (H * W))].contiguous()
            line["cpu_baseline"], dense_idx = cpu_baseline(zs.cpu(), E0.cpu())
            try:
                line["parity"] = parity_gate(zs, E0, dense_idx)
                # the same census over the whole batch against the formulation's GPU form (rocBLAS sgemm +
                # torch.argmin, 64 Ki rows at a time), outside every timed region
                from tools.near_tie_census import census, dense_indices
                got = ops.search(z, E0, ws, want_quantized=False)["indices"]
                full = census(z, E0, got, dense_indices(z, E0))
                full["against"] = "dense torch formulation on the GPU (rocBLAS sgemm + torch.argmin), whole batch"
                line["parity"]["vs_dense_formulation_gpu"] = full
            except Exception as exc:
                line["parity"] = {"error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
