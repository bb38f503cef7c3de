"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares, the
host-only size helpers behave, and the drop-in module keeps the reference's interface.  No
compute call is made here (there is no GPU in this environment)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vq_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = header_symbols()
    for must in ("vq_search", "vq_codebook_prep", "vq_ema_scatter", "vq_ema_cluster_update",
                 "vq_ema_codebook_update", "vq_code_histogram", "vq_pairwise_stats", "vq_index_stats"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from vq_vae_amd import _lib
    lib = _lib.load()
    for s in header_symbols():
        assert hasattr(lib, s), f"libvqhip.so lacks {s}"
    assert set(header_symbols()) == set(_lib.SIGNATURES), "ctypes signature table and header disagree"
    assert lib.vq_version() >= 100


def test_size_helpers():
    from vq_vae_amd import _lib
    lib = _lib.load()
    # image: one stage per 64 codes, 64*D floats + a 1 KiB norms piece
    assert lib.vq_image_bytes(8192, 256) == 128 * (64 * 256 + 256) * 4
    assert lib.vq_image_bytes(100, 32) == 2 * (64 * 32 + 256) * 4
    assert lib.vq_norms_bytes(100) == 128 * 4
    assert lib.vq_search_blocks(4194304) == 16384 and lib.vq_search_blocks(1) == 1
    assert lib.vq_loss_partial_bytes(4096) == (16 + 512 + 1024) * 4     # sweep, shared exact rows, exact kernel
    # the persistent kernel: large batches; every codebook size from 128 channels up, K <= 1024 below; not zero-padded counts
    assert lib.vq_search_uses_persistent_kernel(4194304, 8192, 256, 1, 1024) == 1
    assert lib.vq_search_uses_persistent_kernel(4194304, 8192, 64, 1, 1024) == 0
    assert lib.vq_search_uses_persistent_kernel(1048576, 512, 64, 0, 1) == 1
    assert lib.vq_search_uses_persistent_kernel(1048576, 1000, 48, 1, 1024) == 0
    assert lib.vq_search_uses_persistent_kernel(65536, 8192, 256, 1, 1024) == 0
    assert lib.vq_scatter_slabs(4194304, 8192, 256) >= 2
    assert lib.vq_scatter_slabs(128, 512, 64) == 1
    # codebook split scratch: only for small batches with enough stages
    assert lib.vq_search_split_bytes(4194304, 8192) == 0 and lib.vq_search_split_bytes(8192, 64) == 0
    assert lib.vq_search_split_bytes(8192, 8192) == 8 * 8192 * 8
    assert lib.vq_status_string(0) == b"ok"
    assert lib.vq_status_string(1) == b"invalid argument"


def test_entry_points_reject_bad_arguments_without_a_gpu():
    from vq_vae_amd import _lib
    lib = _lib.load()
    # argument validation happens before any HIP call
    assert lib.vq_codebook_prep(None, 512, 64, None, None, None, None) == 1
    assert lib.vq_search(None, 1, 128, 64, 64, None, None, None, 512, 1.0, 0, None, None, None, None, None, None,
                         None, None, None) == 1
    assert lib.vq_code_histogram(None, 0, 10, 512, None, 0, None, None) == 1
    assert lib.vq_ema_scatter(None, 1, 128, 64, 64, None, 512, 0, 0, None, None, 0, None) == 1


def test_module_interface_matches_reference():
    from vq_vae_amd import EMAVectorQuantizer
    m = EMAVectorQuantizer(num_embeddings=512, embedding_dim=64, commitment_cost=3.0, decay=0.95)
    sd = m.state_dict()
    # names, order, shapes, dtypes of vqvae/ema_vector_quantizer.py:21-31
    assert list(sd.keys()) == ["ema_cluster_size", "ema_w", "usage_count", "last_reset_epoch",
                               "last_low_usage_reset_epoch", "embedding.weight"]
    assert sd["ema_cluster_size"].shape == (512,) and sd["ema_w"].shape == (512, 64)
    assert sd["usage_count"].shape == (512,) and sd["last_reset_epoch"].shape == (1,)
    assert sd["embedding.weight"].shape == (512, 64)
    assert all(v.dtype == torch.float32 for v in sd.values())
    assert float(sd["ema_w"].abs().sum()) == 0.0                       # zeros, not a copy of the codebook
    w = sd["embedding.weight"]
    assert float(w.abs().max()) <= 1.0 / 512 and float(w.abs().max()) > 0  # U(-1/K, 1/K)
    assert [n for n, _ in m.named_parameters()] == ["embedding.weight"]
    assert m.embedding.weight.requires_grad
    assert (m.num_embeddings, m.embedding_dim, m.commitment_cost, m.decay, m.epsilon) == (512, 64, 3.0, 0.95, 1e-5)
    for meth in ("expand_codebook", "reset_dead_codes", "get_codebook_stats", "reset_low_usage_codes"):
        assert callable(getattr(m, meth))
    # strict round trip through a second instance
    m2 = EMAVectorQuantizer(512, 64, 3.0)
    m2.load_state_dict(sd, strict=True)
    assert torch.equal(m2.embedding.weight, m.embedding.weight)
    # maintenance methods return early exactly like the reference when not training / no epoch
    m.eval()
    assert m.reset_dead_codes(current_epoch=10) == 0 and m.reset_low_usage_codes(current_epoch=10) == 0
    m.train()
    assert m.reset_dead_codes(current_epoch=None) == 0
    assert m.expand_codebook(512) is False


def test_module_refuses_cpu_tensors():
    from vq_vae_amd import EMAVectorQuantizer
    m = EMAVectorQuantizer(512, 64, 3.0)
    with pytest.raises(RuntimeError, match="only on the GPU"):
        m(torch.randn(2, 64, 4, 4))


def test_expand_codebook_on_cpu_state():
    """expand_codebook is pure state surgery (vqvae/ema_vector_quantizer.py:110-163): old rows kept,
    EMA buffers zero-extended, state_dict still has the six entries."""
    from vq_vae_amd import EMAVectorQuantizer
    torch.manual_seed(0)
    m = EMAVectorQuantizer(64, 16, 3.0)
    m.usage_count[:8] = torch.arange(8, 0, -1).float()
    m.ema_cluster_size[:] = 1.5
    old = m.embedding.weight.detach().clone()
    assert m.expand_codebook(80) is True
    assert m.num_embeddings == 80 and m.embedding.weight.shape == (80, 16)
    assert torch.equal(m.embedding.weight[:64], old)
    assert m.ema_cluster_size.shape == (80,) and float(m.ema_cluster_size[64:].abs().sum()) == 0.0
    assert m.ema_w.shape == (80, 16) and m.usage_count.shape == (80,)
    assert len(m.state_dict()) == 6


def test_ema_scalars_round_like_the_reference_expressions():
    from vq_vae_amd.ops import ema_scalars
    d, omd, eps, keps = ema_scalars(8192, 0.95, 1e-5)
    assert d == float(np.float32(0.95)) and omd == float(np.float32(1 - 0.95))
    assert eps == float(np.float32(1e-5)) and keps == float(np.float32(8192 * 1e-5))


def test_lazy_onehot_metadata():
    from vq_vae_amd import LazyOneHot
    lo = LazyOneHot(torch.tensor([3, 1, 2]), 8)
    assert lo.shape == (3, 8) and lo.dtype == torch.float32 and lo.dim() == 2 and lo.size(1) == 8


def test_checkpoint_helpers_round_trip():
    """A reference-shaped checkpoint (vqgan_state_dict with the quantize.* entries) loads into the
    drop-in module; a row-sharded state reassembles to the same six tensors."""
    from vq_vae_amd import EMAVectorQuantizer, checkpoint
    torch.manual_seed(1)
    src = EMAVectorQuantizer(100, 16, 3.0)
    src.ema_cluster_size.uniform_(0, 3); src.ema_w.normal_(); src.usage_count.uniform_(0, 9)
    src.last_reset_epoch.fill_(6)
    model_state = {"encoder.conv.weight": torch.zeros(3), **{"quantize." + k: v for k, v in src.state_dict().items()}}
    assert checkpoint.infer_shape(model_state) == (100, 16)
    assert checkpoint.model_config(model_state) == {"vq_num_embed": 100, "vq_embed_dim": 16}
    qs = checkpoint.quantizer_state(model_state)
    assert list(qs.keys()) == list(src.state_dict().keys())
    dst = EMAVectorQuantizer(100, 16, 3.0)
    dst.load_state_dict(qs, strict=True)
    assert torch.equal(dst.ema_w, src.ema_w) and float(dst.last_reset_epoch) == 6
    shards = [checkpoint.shard_state(qs, r, 8) for r in range(8)]
    assert sum(s["embedding.weight"].shape[0] for s in shards) == 100
    merged = checkpoint.merge_shards(shards)
    for k in qs:
        assert torch.equal(merged[k], qs[k]), k
    with pytest.raises(KeyError):
        checkpoint.quantizer_state({"quantize.ema_w": torch.zeros(1)})


def test_stats_json_schema():
    from vq_vae_amd.stats import stats_to_json_dict
    res = {"n_embed": 4, "unique_indices": np.array([0, 2]), "utilization": 0.5, "entropy": 1.0,
           "normalized_entropy": 0.5, "gini": 0.0, "index_counts": np.array([3.0, 0, 3, 0]),
           "probs": np.array([0.5, 0, 0.5, 0])}
    d = stats_to_json_dict(res)
    # key set and order of vqvae/codebook_utilization_check.py:208-217
    assert list(d.keys()) == ["n_embed", "unique_indices", "utilization", "entropy", "normalized_entropy", "gini",
                              "index_counts", "probs"]
    import json
    json.dumps(d)


def test_filtered_entry_argument_checks():
    """vq_search_filtered rejects bad arguments and unsupported shapes before any GPU work
    (status codes only; no device is touched on these paths)."""
    from vq_vae_amd import _lib
    lib = _lib.load()
    p = ctypes.c_void_p(0x1000)          # never dereferenced on the paths below
    nul = ctypes.c_void_p(0)

    def call(z=p, N=1024, D=64, K=512, T=1.0, layout=0, hw=1, q=nul, code_base=0):
        return lib.vq_search_filtered(z, layout, N, D, hw, p, p, p, p, K, T, code_base, p, p, nul, q, nul, p, p, p, nul)

    assert call(z=nul) != 0 and b"invalid" in lib.vq_status_string(call(z=nul))
    assert call(N=0) == call(z=nul)
    assert call(layout=7) == call(z=nul)
    assert call(layout=1, hw=100) == call(z=nul)                     # N not a multiple of H*W
    unsupported = call(K=16384)
    assert unsupported != 0 and unsupported != call(z=nul)
    assert call(D=32) == unsupported and call(T=0.0) == unsupported and call(T=float("inf")) == unsupported
    assert call(q=p, code_base=512) == unsupported
    # 168 bytes per row (rounded up to 4 rows); batches of fewer than 192 x 512 rows add the key lists of a
    # divided first pass (32 bytes per row and part, up to 8 parts)
    assert lib.vq_filter_work_bytes(1 << 20) == 168 << 20 and lib.vq_filter_work_bytes((1 << 20) + 1) == 168 * ((1 << 20) + 4)
    assert lib.vq_filter_work_bytes(1000) == (168 + 32 * 8) * 1000 and lib.vq_filter_work_bytes(65536) == (168 + 32 * 2) * 65536
    assert lib.vq_filter_image_bytes(8192, 256) == 256 * 17 * 1024 + 256 + 8192 * 32 + 256 * 4 * 33
    assert lib.vq_filter_loss_partial_bytes(1 << 20) > lib.vq_loss_partial_bytes(1 << 20)


def test_boundary_module_host_side():
    """QuantizerBoundary: constructor check of vqvae/custom_vqgan.py:25-26, the reference's attribute
    names, loading of a CustomVQGAN-style state dict, refusal of CPU tensors."""
    import torch
    from vq_vae_amd import QuantizerBoundary
    with pytest.raises(ValueError):
        QuantizerBoundary(latent_channels=32, n_embed=16, embed_dim=64)
    b = QuantizerBoundary(latent_channels=8, n_embed=16, embed_dim=8, ema_decay=0.9, commitment_loss_beta=2.0)
    assert isinstance(b.quant_conv, torch.nn.Conv2d) and b.quant_conv.kernel_size == (1, 1)
    assert isinstance(b.post_quant_conv, torch.nn.Conv2d) and b.post_quant_conv.out_channels == 8
    assert b.quantize.num_embeddings == 16 and b.quantize.decay == 0.9 and b.quantize.commitment_cost == 2.0
    assert b.config == {"num_vq_embeddings": 16, "vq_embed_dim": 8, "ema_decay": 0.9, "commitment_loss_beta": 2.0}
    names = set(b.state_dict())
    assert {"quant_conv.weight", "quant_conv.bias", "post_quant_conv.weight", "post_quant_conv.bias",
            "quantize.embedding.weight", "quantize.ema_w", "quantize.ema_cluster_size", "quantize.usage_count",
            "quantize.last_reset_epoch", "quantize.last_low_usage_reset_epoch"} == names
    sd = {k: torch.randn_like(v) for k, v in b.state_dict().items()}
    sd["encoder.conv_in.weight"] = torch.zeros(3)
    sd["decoder.conv_out.bias"] = torch.zeros(3)
    b.load_vqgan_state_dict(sd)
    for k, v in b.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # the log-variance half of a 2x-channel encoder output is dropped before quant_conv (:66-67)
    z = b.to_latents(torch.randn(2, 16, 4, 4))
    assert z.shape == (2, 8, 4, 4)
    with pytest.raises(RuntimeError):
        b(torch.randn(2, 8, 4, 4))


def test_filter_sweep_split_fits_the_work_buffer():
    """The divided sweep of the filtered search (host-side rule, no GPU): every part gets at least one
    stage, the first pass's per-part key lists fit behind the fixed regions of the work buffer, and
    the second pass (at most N/8 rows) never holds more than N (row, part) lists."""
    from vq_vae_amd import _lib
    lib = _lib.load()
    parts = (ctypes.c_int * 2)()

    def split(rows, K):
        assert lib.vq_filter_split(rows, K, parts) == 0
        return parts[0], parts[1]

    rng = np.random.default_rng(5)
    Ns = [1, 2, 31, 512, 513, 4096, 4999, 16384, 65536, 98303, 98304, 98305, 131072, 1 << 20, 1 << 22]
    Ns += [int(x) for x in rng.integers(1, 1 << 22, 40)]
    for N in Ns:
        NP = (N + 3) // 4 * 4
        work = lib.vq_filter_work_bytes(N)
        for K in (1, 100, 128, 129, 512, 1000, 2048, 8192):
            stages = (K + 127) // 128
            S, sps = split(N, K)
            assert 1 <= S <= 8 and sps >= 1 and S * sps >= stages and (S - 1) * sps < stages
            if stages >= 2:
                assert sps >= 2 or S == 1 or stages < 2 * S          # at least two stages per part where possible
            wgs = (N + 511) // 512
            assert S == 1 or S * wgs <= 256                            # parts only while the chip is not full
            if S > 1:
                assert 168 * NP + 32 * NP * S <= work                  # first pass: lists of every part
            else:
                assert 168 * NP <= work
            for count in {0, 1, 511, 512, 513, N // 16, N // 8}:
                if count > N // 8:
                    continue
                S2, _ = split(count, K)
                assert count * S2 * 64 <= NP * 64                      # second pass: within its N x 64 byte region
                assert count * 64 <= NP * 8 and count * 32 <= NP * 4   # folded lists and row terms


def test_header_and_library_from_plain_c(tmp_path):
    """include/vq_hip.h compiles as C99 (-pedantic -Werror) and libvqhip.so links and runs from a C
    program that calls the host-only entry points (tests/cabi/link_check.c)."""
    import shutil
    import subprocess
    from vq_vae_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    _lib.load()
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "link_check")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cabi", "link_check.c"), "-L", libdir, "-lvqhip",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "link_check ok" in run.stdout, run.stdout + run.stderr


def test_reference_layout_checkpoint_file_loads(tmp_path):
    """A checkpoint file in the reference trainer's layout (tests/golden/g10_checkpoint_light.pt, written by
    tests/golden/make_golden.py from the reference quantizer after two training steps) is read with
    torch.load(weights_only=True), its six quantizer tensors load strictly into this module, and the
    oracle run from that state reproduces the reference's eval forward stored beside it."""
    import torch
    from conftest import GOLDEN, rel_err
    from oracle import vq_oracle as vo
    from vq_vae_amd import EMAVectorQuantizer, checkpoint
    ck = checkpoint.load_reference_checkpoint(os.path.join(GOLDEN, "g10_checkpoint_light.pt"))
    g = np.load(os.path.join(GOLDEN, "g10_after_load.npz"))
    assert (ck["num_embeddings"], ck["embedding_dim"]) == (64, 16) and ck["epoch"] == 7
    assert ck["model_config"] == {"vq_num_embed": 64, "vq_embed_dim": 16}
    assert set(ck["model_state"]) >= {"quant_conv.weight", "post_quant_conv.bias", "quantize.ema_w"}
    m = EMAVectorQuantizer(ck["num_embeddings"], ck["embedding_dim"], 3.0)
    m.load_state_dict(ck["quantizer_state"], strict=True)
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), g["state_" + k.replace(".", "_")]), k
    assert int(m.last_reset_epoch) == 3
    st = vo.OracleState(embedding=m.embedding.weight.detach().numpy().copy(), ema_w=m.ema_w.numpy().copy(),
                        ema_cluster_size=m.ema_cluster_size.numpy().copy(), usage_count=m.usage_count.numpy().copy())
    out = vo.forward(g["z"], st, 3.0, 0.95, 1e-5, 1.0, False)
    assert np.array_equal(out["indices"].reshape(-1), g["indices"].reshape(-1))
    assert rel_err(out["quantized"], g["quantized"]) <= 1e-6 and abs(out["loss"] - g["loss"]) <= 1e-5 * g["loss"]
    # and back: the light variant written here is read by the same loader
    full_state = dict(ck["model_state"])
    path = tmp_path / "resaved_best_light.pt"
    torch.save(checkpoint.light_checkpoint(full_state, epoch=8, best_val_loss=0.5, codebook_stats=ck["codebook_stats"]),
               str(path))
    again = checkpoint.load_reference_checkpoint(str(path))
    assert again["epoch"] == 8 and again["model_config"] == ck["model_config"]
    for k in ck["quantizer_state"]:
        assert torch.equal(again["quantizer_state"][k], ck["quantizer_state"][k])


def test_light_checkpoint_holds_what_the_trainer_loader_reads(tmp_path):
    """The same key accesses as the reference trainer's loader (vqvae/vqgan_trainer.py:474-526 with
    skip_optimizer): model_config['vq_num_embed'], 'vqgan_state_dict', 'discriminator_state_dict' (indexed,
    not .get), the optional scheduler states, 'codebook_stats', 'epoch', 'best_val_loss' -- and none of the
    optimizer / scaler entries the light variant drops (:442)."""
    import torch
    from vq_vae_amd import EMAVectorQuantizer, checkpoint
    m = EMAVectorQuantizer(32, 16, 3.0)
    model_state = {"quantize." + k: v for k, v in m.state_dict().items()}
    disc = {"model.0.weight": torch.randn(4, 3, 4, 4), "model.0.bias": torch.zeros(4)}
    sched = {"last_epoch": 5, "_step_count": 6, "base_lrs": [1e-4]}
    path = tmp_path / "vqgan_best_light.pt"
    torch.save(checkpoint.light_checkpoint(model_state, epoch=5, best_val_loss=0.7, codebook_stats=[{"epoch": 5}],
                                           discriminator_state=disc, g_scheduler_state=sched, d_scheduler_state=sched),
               str(path))
    ck = torch.load(str(path), map_location="cpu", weights_only=True)
    assert ck["model_config"]["vq_num_embed"] == 32 and ck["model_config"]["vq_embed_dim"] == 16   # :478-487
    m2 = EMAVectorQuantizer(32, 16, 3.0)
    m2.load_state_dict({k[len("quantize."):]: v for k, v in ck["vqgan_state_dict"].items()}, strict=True)   # :491
    assert set(ck["discriminator_state_dict"]) == set(disc)                                        # :492
    assert ck["g_scheduler_state_dict"]["last_epoch"] == 5 and "d_scheduler_state_dict" in ck       # :501-504
    assert ck["codebook_stats"] == [{"epoch": 5}] and ck.get("epoch", 0) + 1 == 6                   # :517-521
    assert ck.get("best_val_loss", 0.0) == 0.7                                                      # :523
    assert not [k for k in ck if "optimizer" in k or "scaler" in k]                                 # :442
    # without a discriminator state the file serves the utilisation checker only: the key is absent
    assert "discriminator_state_dict" not in checkpoint.light_checkpoint(model_state, epoch=1)
