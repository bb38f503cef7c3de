#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by running the REFERENCE quantizer.

Run in the build container only (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference module (vqvae/ema_vector_quantizer.py) is imported from /root/reference and
executed on CPU; nothing of its source is stored, only inputs and outputs.  All inputs come
from seeded NumPy PCG64 streams (helpers in tests/golden/inputs.py) so the tests can
regenerate the large ones instead of storing them.

Two more reference functions run here although their modules cannot be imported in this image
(vqvae/codebook_utilization_check.py and vqvae/vqgan_trainer.py import torchvision / diffusers at
module level): analyze_codebook_utilization and VQGANTrainer.monitor_codebook use neither, so
`reference_function` compiles just that one definition out of the reference file's syntax tree
(read from /root/reference at run time) and g12 / g13 / g14 (VQModelTrainer.get_perplexity) call it with small stand-in *arguments* (a
model object that returns given indices; a trainer namespace around the real reference quantizer).
CustomVQGAN's constructor needs diffusers' encoder / decoder, but its encode / decode / forward
methods only call them: g15 binds those three methods to a module with identity encoder / decoder.
"""
import ast
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from inputs import gauss, ref_init_codebook  # noqa: E402
from vqvae.ema_vector_quantizer import EMAVectorQuantizer  # noqa: E402  (the reference)

torch.set_num_threads(4)


def make_ref(K, D, beta, decay, E):
    q = EMAVectorQuantizer(K, D, beta, decay=decay)
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(E))
    return q


def state_of(q):
    sd = q.state_dict()
    return {f"state_{k.replace('.', '_')}": v.detach().numpy().copy() for k, v in sd.items()}


def run_step(q, z, T):
    out = q(torch.from_numpy(z), temperature=T)
    quant, loss, (perp, idx, enc) = out
    assert enc.shape == (idx.shape[0], q.num_embeddings)
    return {"quantized": quant.detach().numpy().copy(), "loss": np.float32(loss.item()),
            "perplexity": np.float32(perp.item()), "indices": idx.numpy().astype(np.int64).copy()}


def g1():
    """Single step, K=512 D=64 N=128, train/eval, T in {1, 2.5}, two codebook inits."""
    K, D = 512, 64
    z = gauss(101, (2, D, 8, 8))
    for init in ("gauss", "refinit"):
        E = gauss(102, (K, D)) * np.float32(0.5) if init == "gauss" else ref_init_codebook(103, K, D)
        for mode in ("train", "eval"):
            for T in (1.0, 2.5):
                q = make_ref(K, D, 3.0, 0.95, E)
                q.train(mode == "train")
                o = run_step(q, z, T)
                stats = q.get_codebook_stats()
                rec = {"z_seed": 101, "E": E, "T": np.float32(T), "beta": np.float32(3.0), "decay": np.float64(0.95),
                       "training": np.bool_(mode == "train"), **o, **state_of(q),
                       "stats": np.array([stats["active_size"], stats["utilization"], stats["entropy"],
                                          stats["normalized_entropy"], stats["total_usage"]], np.float64)}
                np.savez_compressed(os.path.join(HERE, f"g1_{init}_{mode}_T{T}.npz"), **rec)


def g2():
    """5-step EMA trajectory, fresh batch each step; pins the persisted Laplace smoothing."""
    K, D = 128, 32
    E0 = gauss(202, (K, D)) * np.float32(0.7)
    for decay, beta, T in ((0.95, 3.0, 1.0), (0.7, 10.0, 2.5)):
        q = make_ref(K, D, beta, decay, E0)
        q.train(True)
        rec = {"E0": E0, "decay": np.float64(decay), "beta": np.float32(beta), "T": np.float32(T),
               "z_seeds": np.arange(210, 215)}
        for step in range(5):
            z = gauss(210 + step, (4, D, 8, 8))
            o = run_step(q, z, T)
            for k, v in o.items():
                if k != "quantized":
                    rec[f"s{step}_{k}"] = v
            rec[f"s{step}_quantized_sum"] = np.float64(o["quantized"].astype(np.float64).sum())
            for k, v in state_of(q).items():
                rec[f"s{step}_{k}"] = v
        np.savez_compressed(os.path.join(HERE, f"g2_traj_decay{decay}.npz"), **rec)


def g3():
    """Engineered ties: duplicated codebook rows and latents equidistant from two codes."""
    K, D = 16, 8
    E = np.zeros((K, D), np.float32)
    E[:, 0] = np.arange(K, dtype=np.float32)          # codes on a line at 0,1,2,...
    E[5] = E[3]                                         # exact duplicate of code 3
    E[12] = E[3]
    z = np.zeros((1, D, 4, 4), np.float32)
    vals = np.array([3.0, 0.5, 1.5, 2.5, 7.5, 14.5, 100.0, -4.0, 3.0, 6.5, 9.5, 10.5, 0.0, 15.0, 8.0, 11.5], np.float32)
    z[0, 0] = vals.reshape(4, 4)
    q = make_ref(K, D, 1.0, 0.95, E)
    q.eval()
    o = run_step(q, z, 1.0)
    np.savez_compressed(os.path.join(HERE, "g3_ties.npz"), E=E, z=z, **o)
    # NaN / inf handling of torch.argmin on this path
    z2 = z.copy()
    z2[0, 1, 0, 0] = np.nan
    z2[0, 2, 0, 1] = np.inf
    z2[0, 3, 0, 2] = -np.inf
    q = make_ref(K, D, 1.0, 0.95, E)
    q.eval()
    o = run_step(q, z2, 1.0)
    np.savez_compressed(os.path.join(HERE, "g3_nonfinite.npz"), E=E, z=z2, indices=o["indices"])
    # negative / zero temperature flips or degenerates the order (literal division, :49-50)
    for T in (-1.0, 0.0):
        q = make_ref(K, D, 1.0, 0.95, E)
        q.eval()
        o = run_step(q, z, T)
        np.savez_compressed(os.path.join(HERE, f"g3_T{T}.npz"), E=E, z=z, T=np.float32(T), indices=o["indices"])


def g4():
    """Backward: d(a*sum(q*w) + b*loss)/dz for the straight-through + commitment term."""
    K, D = 64, 16
    E = gauss(402, (K, D)) * np.float32(0.6)
    z = gauss(401, (2, D, 4, 4))
    w = gauss(403, (2, D, 4, 4))
    for mode in ("train", "eval"):
        q = make_ref(K, D, 3.0, 0.95, E)
        q.train(mode == "train")
        zt = torch.from_numpy(z).requires_grad_(True)
        quant, loss, _ = q(zt)
        obj = 0.7 * (quant * torch.from_numpy(w)).sum() + 1.3 * loss
        obj.backward()
        np.savez_compressed(os.path.join(HERE, f"g4_backward_{mode}.npz"), E=E, z=z, w=w, a=np.float32(0.7),
                            b=np.float32(1.3), beta=np.float32(3.0), grad=zt.grad.numpy().copy(),
                            loss=np.float32(loss.item()))


def g6():
    """Full benchmark configuration 1 (K=512, D=64, batch 16 of 16x16 latents): indices stored whole,
    float outputs as float64 checksums; inputs are regenerated from their seeds."""
    K, D = 512, 64
    z = gauss(601, (16, D, 16, 16))
    E = gauss(602, (K, D)) * np.float32(0.5)
    q = make_ref(K, D, 3.0, 0.95, E)
    q.train(True)
    o = run_step(q, z, 1.0)
    st = state_of(q)
    rec = {"z_seed": 601, "E_seed": 602, "E_scale": np.float32(0.5), "indices": o["indices"].astype(np.int16),
           "indices_sha256": hashlib.sha256(o["indices"].astype(np.int64).tobytes()).hexdigest(),
           "loss": o["loss"], "perplexity": o["perplexity"],
           "quantized_sum": np.float64(o["quantized"].astype(np.float64).sum()),
           "quantized_abs_sum": np.float64(np.abs(o["quantized"].astype(np.float64)).sum())}
    for k, v in st.items():
        if v.ndim == 2:
            rec[k + "_sum"] = np.float64(v.astype(np.float64).sum())
            rec[k + "_abs_sum"] = np.float64(np.abs(v.astype(np.float64)).sum())
            rec[k + "_rows"] = v[:8].copy()
        else:
            rec[k] = v
    np.savez_compressed(os.path.join(HERE, "g6_config1.npz"), **rec)


def g8():
    """Headline codebook shape (K=8192, D=256) on a small batch (N=512): indices, loss,
    perplexity; inputs regenerated from seeds."""
    K, D = 8192, 256
    z = gauss(801, (2, D, 16, 16))
    E = gauss(802, (K, D))
    q = make_ref(K, D, 3.0, 0.95, E)
    q.train(True)
    o = run_step(q, z, 1.0)
    st = state_of(q)
    np.savez_compressed(os.path.join(HERE, "g8_k8192.npz"), z_seed=801, E_seed=802,
                        indices=o["indices"].astype(np.int16), loss=o["loss"], perplexity=o["perplexity"],
                        quantized_sum=np.float64(o["quantized"].astype(np.float64).sum()),
                        state_ema_cluster_size=st["state_ema_cluster_size"],
                        state_usage_count=st["state_usage_count"],
                        embedding_sum=np.float64(st["state_embedding_weight"].astype(np.float64).sum()),
                        embedding_abs_sum=np.float64(np.abs(st["state_embedding_weight"].astype(np.float64)).sum()))


def g9():
    """The quantizer between its two 1x1 convolutions, composed as vqvae/custom_vqgan.py does (:50-51
    the layers, :68 quant_conv, :81 quantize, :72 post_quant_conv) from torch's own nn.Conv2d and the
    reference quantizer (CustomVQGAN itself needs diffusers): forward outputs, post-step state and the
    gradients of a * sum(out * w) + b * loss w.r.t. the encoder output and all four conv tensors."""
    K, D = 64, 16
    E = gauss(902, (K, D)) * np.float32(0.6)
    h = gauss(901, (2, D, 4, 4))
    w = gauss(903, (2, D, 4, 4))
    Wq = gauss(904, (D, D, 1, 1)) * np.float32(0.3)
    bq = gauss(905, (D,)) * np.float32(0.1)
    Wp = gauss(906, (D, D, 1, 1)) * np.float32(0.3)
    bp = gauss(907, (D,)) * np.float32(0.1)
    for mode in ("train", "eval"):
        q = make_ref(K, D, 3.0, 0.95, E)
        q.train(mode == "train")
        quant_conv = torch.nn.Conv2d(D, D, 1)
        post_quant_conv = torch.nn.Conv2d(D, D, 1)
        with torch.no_grad():
            quant_conv.weight.copy_(torch.from_numpy(Wq)); quant_conv.bias.copy_(torch.from_numpy(bq))
            post_quant_conv.weight.copy_(torch.from_numpy(Wp)); post_quant_conv.bias.copy_(torch.from_numpy(bp))
        ht = torch.from_numpy(h).requires_grad_(True)
        quant, loss, (perp, idx, _enc) = q(quant_conv(ht))
        out = post_quant_conv(quant)
        obj = 0.7 * (out * torch.from_numpy(w)).sum() + 1.3 * loss
        obj.backward()
        np.savez_compressed(os.path.join(HERE, f"g9_boundary_{mode}.npz"), E=E, h=h, w=w, Wq=Wq, bq=bq, Wp=Wp, bp=bp,
                            a=np.float32(0.7), b=np.float32(1.3), beta=np.float32(3.0), decay=np.float64(0.95),
                            out=out.detach().numpy().copy(), loss=np.float32(loss.item()),
                            perplexity=np.float32(perp.item()), indices=idx.numpy().astype(np.int64).copy(),
                            grad_h=ht.grad.numpy().copy(), grad_Wq=quant_conv.weight.grad.numpy().copy(),
                            grad_bq=quant_conv.bias.grad.numpy().copy(),
                            grad_Wp=post_quant_conv.weight.grad.numpy().copy(),
                            grad_bp=post_quant_conv.bias.grad.numpy().copy(), **state_of(q))


def g10():
    """A checkpoint file in the layout VQGANTrainer.save_checkpoint writes for its light variant
    (vqvae/vqgan_trainer.py:409-423, :442-445): the state_dict of a module tree named like CustomVQGAN's
    (quant_conv, quantize = the reference quantizer after two training steps, post_quant_conv;
    vqvae/custom_vqgan.py:36-51) under 'vqgan_state_dict', plus epoch, best_val_loss, codebook_stats and
    model_config.  Beside it: a batch and the reference's eval-mode outputs after the two steps."""
    K, D = 64, 16
    E = gauss(1002, (K, D)) * np.float32(0.6)
    q = make_ref(K, D, 3.0, 0.95, E)
    q.train(True)
    for step in range(2):
        run_step(q, gauss(1010 + step, (2, D, 4, 4)), 1.0)
    q.last_reset_epoch.fill_(3)
    tree = torch.nn.Module()
    tree.quant_conv = torch.nn.Conv2d(D, D, 1)
    tree.quantize = q
    tree.post_quant_conv = torch.nn.Conv2d(D, D, 1)
    stats = q.get_codebook_stats()
    ckpt = {"epoch": 7, "vqgan_state_dict": tree.state_dict(),
            "discriminator_state_dict": {"main.0.weight": torch.zeros(2, 2)},
            "best_val_loss": float(stats["normalized_entropy"]), "codebook_stats": [dict(stats, epoch=7)],
            "model_config": {"vq_num_embed": q.num_embeddings, "vq_embed_dim": D}}
    torch.save(ckpt, os.path.join(HERE, "g10_checkpoint_light.pt"))
    q.eval()
    z = gauss(1020, (2, D, 4, 4))
    o = run_step(q, z, 1.0)
    np.savez_compressed(os.path.join(HERE, "g10_after_load.npz"), z=z, **o, **state_of(q),
                        stats=np.array([stats["active_size"], stats["utilization"], stats["entropy"],
                                        stats["normalized_entropy"], stats["total_usage"]], np.float64))


def g11():
    """The call pattern of VQGANTrainer on the quantizer (vqvae/vqgan_trainer.py): per epoch
    reset_dead_codes(threshold=2, current_epoch) in training mode (:339; nothing is dead enough in
    this data, so it returns 0 -- the return value is stored), then per batch one forward under
    torch.no_grad() (discriminator step, :86) and one with gradients (generator step, :136) --
    both in training mode, so the EMA update runs twice per batch -- a backward through
    quantized and loss, and get_codebook_stats() as CustomVQGAN.forward(return_dict=True) calls it
    (vqvae/custom_vqgan.py:102).  T = 2.5 as in vqvae/improved_vq_config.py:60."""
    K, D, T = 64, 16, 2.5
    E = gauss(1102, (K, D)) * np.float32(0.6)
    q = make_ref(K, D, 10.0, 0.7, E)
    q.train(True)
    rec = {"E0": E, "T": np.float32(T), "beta": np.float32(10.0), "decay": np.float64(0.7)}
    step = 0
    for epoch in (1, 2):
        rec[f"e{epoch}_reset"] = np.int64(q.reset_dead_codes(threshold=2, current_epoch=epoch))
        for b in range(2):
            z = gauss(1110 + step, (2, D, 4, 4))
            w = gauss(1130 + step, (2, D, 4, 4))
            with torch.no_grad():
                o1 = run_step(q, z, T)
            zt = torch.from_numpy(z).requires_grad_(True)
            quant, loss, (perp, idx, _enc) = q(zt, temperature=T)
            ((quant * torch.from_numpy(w)).sum() + loss).backward()
            st = q.get_codebook_stats()
            rec[f"s{step}_idx_nograd"] = o1["indices"]
            rec[f"s{step}_idx"] = idx.numpy().astype(np.int64).copy()
            rec[f"s{step}_loss"] = np.float32(loss.item())
            rec[f"s{step}_perplexity"] = np.float32(perp.item())
            rec[f"s{step}_grad"] = zt.grad.numpy().copy()
            rec[f"s{step}_stats"] = np.array([st["active_size"], st["utilization"], st["entropy"],
                                              st["normalized_entropy"], st["total_usage"]], np.float64)
            for k, v in state_of(q).items():
                rec[f"s{step}_{k}"] = v
            step += 1
    np.savez_compressed(os.path.join(HERE, "g11_trainer_pattern.npz"), **rec)


def reference_function(rel_path, name, namespace):
    """The function (or method) `name` of the reference file, compiled on its own from the file's
    syntax tree and bound in `namespace` (which supplies the globals its body uses)."""
    path = os.path.join("/root/reference", rel_path)
    with open(path, encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), namespace)
            return namespace[name]
    raise KeyError(name)


def locals_at_return(fn, *args, **kwargs):
    """Call fn and return (result, its local variables at the moment it returned)."""
    seen = {}

    def prof(frame, event, arg):
        if event == "return" and frame.f_code is fn.__code__:
            seen.update(frame.f_locals)

    sys.setprofile(prof)
    try:
        out = fn(*args, **kwargs)
    finally:
        sys.setprofile(None)
    return out, seen


def index_cases():
    """Index streams for the statistics: (name, K, int64 indices)."""
    rng = np.random.Generator(np.random.PCG64(1200))
    cases = [("uniform", 64, rng.integers(0, 64, 5000)),
             ("single", 64, np.full(777, 13)),
             ("all_once", 64, rng.permutation(64)),
             ("half", 64, rng.integers(0, 32, 3000)),
             ("wide", 512, rng.integers(0, 512, 4096))]
    w = 1.0 / np.arange(1, 65) ** 1.1
    cases.append(("zipf", 64, rng.choice(64, 6000, p=w / w.sum())))
    return [(n, k, np.asarray(i, np.int64)) for n, k, i in cases]


def g12():
    """analyze_codebook_utilization (vqvae/codebook_utilization_check.py:64-120), the reference's own
    function, on index streams handed to it by a stand-in model: its "indices" output is whatever the
    batch holds, so the function's statistics (:83-108) run on known indices."""
    from tqdm import tqdm
    fn = reference_function("vqvae/codebook_utilization_check.py", "analyze_codebook_utilization",
                            {"torch": torch, "np": np, "tqdm": lambda it, **kw: it})
    del tqdm

    class Model:
        def __init__(self, K):
            self.quantize = types.SimpleNamespace(num_embeddings=K)

        def eval(self):
            return self

        def __call__(self, images, return_dict=True, temperature=1.0):
            return {"indices": images}

    rec = {}
    for name, K, idx in index_cases():
        batches = [(torch.from_numpy(part), None) for part in np.array_split(idx, 3)]
        r = fn(Model(K), batches, torch.device("cpu"))
        rec[f"{name}_K"] = np.int64(K)
        rec[f"{name}_indices"] = idx
        rec[f"{name}_unique"] = np.asarray(r["unique_indices"], np.int64)
        rec[f"{name}_scalars"] = np.array([r["utilization"], r["entropy"], r["normalized_entropy"], r["gini"]],
                                          np.float64)
        rec[f"{name}_index_counts"] = np.asarray(r["index_counts"], np.float64)
        rec[f"{name}_probs"] = np.asarray(r["probs"], np.float64)
    np.savez_compressed(os.path.join(HERE, "g12_utilization_reference.npz"), **rec)


def g13():
    """VQGANTrainer.monitor_codebook (vqvae/vqgan_trainer.py:577-650), the reference's own method, called
    on a namespace that holds what it reads: the reference quantizer after three training steps, the
    last batch's indices and the trainer's settings.  Stored: the quantizer's state before the call,
    the values the method computed (its locals when it returned) and what it decided -- the contents
    of re-seeded / new rows are random in the reference and are not stored."""
    fn = reference_function("vqvae/vqgan_trainer.py", "monitor_codebook", {"torch": torch, "print": lambda *a, **k: None})
    K, D = 64, 16
    rec = {}
    cases = [("plain", 0, True, 0.8, 512, 0.1, 0.6),         # epoch 0: statistics only
             ("expand", 7, False, 0.5, 96, 0.1, 0.6),        # utilisation above the threshold: 64 -> 80, then 8 re-seeded
             ("reset", 6, True, 0.8, 512, 0.25, 0.6),        # five epochs since the last re-seeding: 16 codes
             ("capped", 2, False, 0.5, 70, 0.25, 0.6),       # expansion capped by the maximum size: 64 -> 70
             ("collapsed", 4, False, 0.8, 512, 0.1, 0.02)]   # a codebook the data does not use: no expansion
    for name, epoch, disabled, thr, max_size, pct, scale in cases:
        E = gauss(1300, (K, D)) * np.float32(scale)
        q = make_ref(K, D, 3.0, 0.95, E)
        q.train(True)
        idx = None
        for step in range(3):
            z = gauss(1310 + step, (2, D, 8, 8))
            with torch.no_grad():
                idx = run_step(q, z, 1.0)["indices"]
        for k, v in state_of(q).items():
            rec[f"{name}_{k}"] = v
        trainer = types.SimpleNamespace(
            vqgan=types.SimpleNamespace(quantize=q), last_batch_indices=torch.from_numpy(idx).flatten(),
            disable_codebook_expansion=disabled, expansion_threshold=thr, max_codebook_size=max_size,
            reset_low_usage_percentage=pct, _update_optimizers_after_expansion=lambda: None)
        torch.manual_seed(5)
        out, loc = locals_at_return(fn, trainer, epoch)
        rec[f"{name}_args"] = np.array([epoch, int(disabled), thr, max_size, pct], np.float64)
        rec[f"{name}_last_batch_indices"] = idx.reshape(-1)
        rec[f"{name}_returned"] = np.float64(out)
        rec[f"{name}_batch"] = np.array([len(loc["unique_indices"]), loc["real_utilization"], loc["batch_entropy"],
                                         loc["batch_norm_entropy"]], np.float64)
        st = loc["stats"]
        rec[f"{name}_cumulative"] = np.array([st["active_size"], st["utilization"], st["entropy"],
                                              st["normalized_entropy"], st["total_usage"]], np.float64)
        rec[f"{name}_size_after"] = np.int64(q.num_embeddings)
        rec[f"{name}_reset_count"] = np.int64(loc.get("reset_count", 0))
        rec[f"{name}_last_reset_epoch_after"] = np.int64(q.last_low_usage_reset_epoch.item())
        rec[f"{name}_state_shapes_after"] = np.array([q.embedding.weight.shape[0], q.ema_w.shape[0],
                                                      q.ema_cluster_size.shape[0], q.usage_count.shape[0]], np.int64)
    np.savez_compressed(os.path.join(HERE, "g13_monitor_codebook_reference.npz"), **rec)


def g14():
    """VQModelTrainer.get_perplexity (vqvae/trainers/vqtrainer.py:46-82), the reference's own method: the
    number of distinct nearest codes of an encoder output's latents.  The trainer and the encoder output
    are namespaces holding what the method reads (model.quantize.embedding / .vq_embed_dim, .latents)."""
    fn = reference_function("vqvae/trainers/vqtrainer.py", "get_perplexity", {"torch": torch})
    rec = {}
    for name, K, D, shape, scale, half in (("small", 64, 16, (2, 16, 8, 8), 0.6, False),
                                           ("wide", 512, 64, (3, 64, 8, 8), 1.0, False),
                                           ("collapsed", 128, 32, (2, 32, 8, 8), 0.01, False),
                                           ("half", 64, 16, (2, 16, 8, 8), 0.6, True)):
        E = gauss(1400, (K, D)) * np.float32(scale)
        z = gauss(1401, shape)
        emb = torch.nn.Embedding(K, D)
        with torch.no_grad():
            emb.weight.copy_(torch.from_numpy(E))
        trainer = types.SimpleNamespace(model=types.SimpleNamespace(quantize=types.SimpleNamespace(embedding=emb, vq_embed_dim=D)))
        lat = torch.from_numpy(z)
        if half:
            lat = lat.half()
            z = lat.float().numpy()          # the values the method works on after its cast (:59)
        out = fn(trainer, types.SimpleNamespace(latents=lat))
        rec[f"{name}_E"] = E
        rec[f"{name}_latents"] = z
        rec[f"{name}_used"] = np.int64(out)
    np.savez_compressed(os.path.join(HERE, "g14_get_perplexity_reference.npz"), **rec)


def g15():
    """CustomVQGAN.encode / .decode / .forward (vqvae/custom_vqgan.py:62-114), the reference's own methods,
    bound to a module that holds what they read: quant_conv and post_quant_conv as nn.Conv2d(D, D, 1),
    the reference quantizer, and identity encoder / decoder (the real ones need diffusers and are not
    on this path).  return_dict=True in training and eval mode, T = 1 and 2.5, and an encoder output of
    2 D channels (the mean / log-variance form, :66-67); gradients of a * sum(decoded * w) + b * loss."""
    ns = {"torch": torch}
    methods = {n: reference_function("vqvae/custom_vqgan.py", n, ns) for n in ("encode", "decode", "forward")}
    Host = type("Host", (torch.nn.Module,), dict(methods))
    K, D = 64, 16
    E = gauss(1502, (K, D)) * np.float32(0.6)
    Wq = gauss(1504, (D, D, 1, 1)) * np.float32(0.3)
    bq = gauss(1505, (D,)) * np.float32(0.1)
    Wp = gauss(1506, (D, D, 1, 1)) * np.float32(0.3)
    bp = gauss(1507, (D,)) * np.float32(0.1)
    w = gauss(1503, (2, D, 4, 4))
    for name, mode, T, channels in (("train", "train", 1.0, D), ("eval", "eval", 1.0, D), ("train_T", "train", 2.5, D),
                                    ("train_2D", "train", 1.0, 2 * D)):
        x = gauss(1501, (2, channels, 4, 4))
        m = Host()
        m.encoder, m.decoder = torch.nn.Identity(), torch.nn.Identity()
        m.quantize = make_ref(K, D, 3.0, 0.95, E)
        m.quant_conv, m.post_quant_conv = torch.nn.Conv2d(D, D, 1), torch.nn.Conv2d(D, D, 1)
        with torch.no_grad():
            m.quant_conv.weight.copy_(torch.from_numpy(Wq)); m.quant_conv.bias.copy_(torch.from_numpy(bq))
            m.post_quant_conv.weight.copy_(torch.from_numpy(Wp)); m.post_quant_conv.bias.copy_(torch.from_numpy(bp))
        m.train(mode == "train")
        xt = torch.from_numpy(x).requires_grad_(True)
        out = m(xt, return_dict=True, temperature=T)
        assert sorted(out) == ["commitment_loss", "decoded_imgs", "entropy", "indices", "normalized_entropy", "perplexity"]
        obj = 0.7 * (out["decoded_imgs"] * torch.from_numpy(w)).sum() + 1.3 * out["commitment_loss"]
        obj.backward()
        tup = m(torch.from_numpy(x), return_dict=False, temperature=T) if mode == "eval" else None
        rec = dict(E=E, x=x, w=w, Wq=Wq, bq=bq, Wp=Wp, bp=bp, T=np.float32(T), a=np.float32(0.7), b=np.float32(1.3),
                   beta=np.float32(3.0), decay=np.float64(0.95),
                   decoded_imgs=out["decoded_imgs"].detach().numpy().copy(),
                   commitment_loss=np.float32(out["commitment_loss"].item()),
                   indices_onehot=out["indices"].detach().numpy().copy(),
                   perplexity=np.float32(out["perplexity"].item()), entropy=np.float32(out["entropy"].item()),
                   normalized_entropy=np.float32(out["normalized_entropy"].item()),
                   grad_x=xt.grad.numpy().copy(), grad_Wq=m.quant_conv.weight.grad.numpy().copy(),
                   grad_bq=m.quant_conv.bias.grad.numpy().copy(), grad_Wp=m.post_quant_conv.weight.grad.numpy().copy(),
                   grad_bp=m.post_quant_conv.bias.grad.numpy().copy(), **state_of(m.quantize))
        if tup is not None:
            rec["tuple_decoded"] = tup[0].detach().numpy().copy()
            rec["tuple_loss"] = np.float32(tup[1].item())
            rec["tuple_perplexity"] = np.float32(tup[2].item())
        np.savez_compressed(os.path.join(HERE, f"g15_vqgan_forward_{name}.npz"), **rec)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g6, g8, g9, g10, g11, g12, g13, g14, g15):
        if not only or fn.__name__ in only:
            fn()
    total = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print("golden bytes:", total)
