#!/usr/bin/env python
"""Headline benchmark: audio-seconds/sec of one zipformer pruned-RNN-T training step.

    python bench.py --gpus N --steps K --warmup W

N>1 without RANK/WORLD_SIZE in the environment: this process starts N ranks itself
(`python -m torch.distributed.run --nproc-per-node N ... bench.py ...`) BEFORE touching the GPU,
relays rank 0's JSON line and exits with the child's code.  Under an external launcher it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (RCCL = backend "nccl").

Workload (BASELINE.json metric, SURVEY.md section 8d "C3"): per-rank batch of 64 synthetic 16 kHz
10 s utterances (PCM resident in HBM before the timed region), 50 labels each from a 500-piece
vocabulary, zipformer YAML dims (config/training/zipformer_stateless_pruned_rnnt.yaml:49-95),
stateless predictor, pruned RNN-T (prune_range 5) with 0.5*simple + 0.5*pruned loss, fp32.
A step = on-GPU fbank -> CMVN -> Zipformer2 fwd -> predictor -> joiner (simple loss, prune
ranges, fused pruned lattice) -> backward -> bucketed RCCL gradient all-reduce (N>1) ->
grad-norm clip 5.0 -> ScaledAdam step -> Eden step.  Weak scaling: per-GPU work is fixed.
`--config C2` times the conformer-CTC step (12 layers, d=256, B=32) instead; the headline is C3.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline / cpu_baseline
objects).
"""
import argparse
import json
import os
import random
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR = 16000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA dense peak
DEFAULT_ROOFLINE_KERNEL = "auto"   # or any entry point declared in include/s2t_mi355.h


def c3_config(vocab=500):
    return {
        "task": {"type": "Pruned_Rnnt", "name": "bench-c3", "export_path": "/tmp"},
        "dataset": {"feat_type": "lhotes_fbank",
                    "feat_config": {"num_mel_bins": 80, "snip_edges": True}},
        "encoder": {"model": "Zipformer", "config": {
            "feature_dim": 80, "downsampling_factor": [1, 2, 4, 8, 4, 2],
            "num_encoder_layers": [2, 2, 2, 2, 2, 2],
            "feedforward_dim": [512, 768, 768, 768, 768, 768],
            "encoder_dim": [192, 256, 256, 256, 256, 256],
            "encoder_unmasked_dim": [192, 192, 192, 192, 192, 192],
            "num_heads": [4, 4, 4, 8, 4, 4], "query_head_dim": 32, "value_head_dim": 12,
            "pos_head_dim": 4, "pos_dim": 48, "cnn_module_kernel": [31, 31, 15, 15, 15, 31],
            "causal": True, "chunk_size": [-1], "left_context_frames": [-1], "for_ctc": False}},
        "decoder": {"model": "Identity", "config": {"dummy": -1}},
        "predictor": {"model": "Stateless", "config": {
            "num_symbols": vocab, "output_dim": 256, "symbol_embedding_dim": 512,
            "context_size": 5}},
        "joiner": {"input_dim": 256, "output_dim": vocab, "prune_range": 5,
                   "use_out_project": False},
        "loss": {"model": "Pruned_Rnnt", "simple_loss_scale": 0.5, "pruned_loss_scale": 0.5,
                 "config": {"termination_symbol": 0, "reduction": "mean"}, "enable_ctc": False},
        "optim_setup": {"seperate_lr": {"apply": False},
                        "optimizer": {"type": "ScaledAdam",
                                      "config": {"lr": 0.045, "clipping_scale": 2.0}},
                        "lr_scheduler": {"type": "Eden", "config": {"lr_batches": 7000},
                                         "step_config": {"interval": "step", "frequency": 1}}},
        "trainer": {"accelerator": "gpu", "devices": 1,
                    "strategy": "ddp_find_unused_parameters_true", "precision": "32-true",
                    "max_epochs": 1, "accumulate_grad_batches": 1, "gradient_clip_val": 5.0,
                    "gradient_clip_algorithm": "norm"},
    }


def c2_config(vocab=128, layers=12):
    """conformer-CTC (config/training/conformer_ctc.yaml dims; BASELINE.json: 12 layers, d=256)."""
    return {
        "task": {"type": "CTC", "name": "bench-c2", "export_path": "/tmp"},
        "dataset": {"feat_type": "fbank", "feat_config": {"num_mel_bins": 80}},
        "encoder": {"model": "Conformer", "config": {
            "bn_cmvn": False, "feats_dim": 80, "subsampling_rate": 4, "input_dim": 256,
            "num_heads": 4, "ffn_dim": 2048, "num_layers": layers,
            "depthwise_conv_kernel_size": 31, "dropout": 0.1, "use_group_norm": False,
            "convolution_first": False, "output_dim": 256}},
        "decoder": {"model": "Projector", "config": {"input_dim": 256, "output_dim": vocab,
                                                     "dropout_p": 0.1}},
        "loss": {"model": "CTC", "config": {"blank_label": 0, "reduction": "mean"}},
        "optim_setup": {"seperate_lr": {"apply": False},
                        "optimizer": {"type": "AdamW", "config": {"lr": 1.0e-3}},
                        "lr_scheduler": {"type": "Warmup", "config": {"warmup_steps": 25000},
                                         "step_config": {"interval": "step", "frequency": 1}}},
        "trainer": {"accelerator": "gpu", "devices": 1, "strategy": "ddp",
                    "precision": "32-true", "max_epochs": 1, "accumulate_grad_batches": 1,
                    "gradient_clip_val": 5.0, "gradient_clip_algorithm": "norm"},
    }


def c4_config(vocab=128, layers=12):
    """CTC_Hybrid_Rnnt (config/training/conformer_hybrid_rnnt.yaml dims): shared conformer encoder
    + CTC head + LSTM predictor + unpruned joiner with the 2-Linear out-projection, 0.8 * rnnt +
    0.2 * ctc (task_factory/rnnt_task.py:304-363)."""
    cfg = c2_config(vocab, layers)
    cfg["task"] = {"type": "CTC_Hybrid_Rnnt", "name": "bench-c4", "export_path": "/tmp"}
    cfg["predictor"] = {"model": "Lstm", "config": {
        "num_symbols": vocab, "output_dim": 256, "symbol_embedding_dim": 256, "num_lstm_layers": 2,
        "lstm_hidden_dim": 256, "lstm_layer_norm": True, "lstm_layer_norm_epsilon": 1e-3,
        "lstm_dropout": 0.3}}
    cfg["joiner"] = {"input_dim": 256, "output_dim": vocab, "inner_dim": 256, "activation": "tanh",
                     "prune_range": -1}
    cfg["loss"] = {"rnnt_weight": 0.8, "ctc_weight": 0.2,
                   "rnnt_loss": {"model": "Rnnt", "config": {"blank_label": 0, "reduction": "mean"}},
                   "ctc_loss": {"model": "CTC", "config": {"blank_label": 0, "reduction": "mean"}}}
    return cfg


def c5_config(layers=12):
    """BEST-RQ SSL pre-training (config/training/conformer_ssl.yaml:45-97): random-projection
    quantizer (720 -> 16, 8192 codes, cosine) + masked conformer + Projector to 8193 classes +
    masked KL loss (task_factory/ssl_task.py:114-180)."""
    cfg = c2_config(128, layers)
    cfg["task"] = {"type": "SSL", "name": "bench-c5", "export_path": "/tmp"}
    cfg["ssl_layer"] = {"model": "Best-RQ",
                        "layer_config": {"cnn_kernel_size": [3, 3], "cnn_stride": [2, 2],
                                         "feat_dim": 80, "num_codebooks": 1, "codebook_dim": 16,
                                         "codebook_size": 8192, "label_basis": "cosine"},
                        "masking_config": {"mask_proportion": 0.5, "mean_span_length": 1,
                                           "span_select_type": "static", "min_num_spans": 1,
                                           "no_overlap": False, "min_space": 0, "seed": 1234}}
    cfg["logits_layer"] = {"model": "Projector", "config": {"input_dim": 256, "output_dim": 8193,
                                                           "dropout_p": 0.0}}
    cfg["loss"] = {"loss_select": "mask_loss", "model": "MaskedKLDiv",
                   "config": {"num_classes": 8193, "scale_factor": 1.0, "label_smoothing": 0.1}}
    cfg.pop("decoder", None)
    return cfg


def synth_pcm(rng, batch, seconds):
    """Band-limited noise + 3 sinusoids, clipped to [-1,1]."""
    n = int(seconds * SR)
    x = 0.1 * rng.standard_normal((batch, n)).astype(np.float32)
    k = np.ones(8, np.float32) / 8.0                      # crude low-pass
    x = np.stack([np.convolve(r, k, mode="same") for r in x])
    t = np.arange(n, dtype=np.float32) / SR
    for _ in range(3):
        f = rng.uniform(100, 4000, size=(batch, 1)).astype(np.float32)
        a = rng.uniform(0.02, 0.2, size=(batch, 1)).astype(np.float32)
        x += a * np.sin(2 * np.pi * f * t[None, :])
    return np.clip(x, -1, 1).astype(np.float32)


def make_batch(rank, batch, seconds, n_labels, vocab, device):
    import torch
    rng = np.random.default_rng(20241218 + rank)
    pcm = synth_pcm(rng, batch, seconds)
    lab = rng.integers(1, vocab - 1, size=(batch, n_labels))
    return {"pcm": torch.from_numpy(pcm).to(device),
            "pcm_length": torch.full((batch,), pcm.shape[1], dtype=torch.int64, device=device),
            "label": torch.from_numpy(lab).to(device),
            "label_length": torch.full((batch,), n_labels, dtype=torch.int64, device=device)}


# ------------------------------------------------------------------ CPU baseline (oracle)
def host_threads(cap=16):
    """Threads this process may really use: affinity mask, cgroup quota, and the GPU box's
    per-GPU CPU share (16) -- oversubscribing a quota-limited container stalls OpenMP."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, min(n, cap))


def _zcfg(ec):
    ns = len(ec["downsampling_factor"])
    tup = lambda v: tuple(v) if isinstance(v, (list, tuple)) else (v,) * ns   # noqa: E731
    return dict(downsampling_factor=tup(ec["downsampling_factor"]),
                num_encoder_layers=tup(ec["num_encoder_layers"]), encoder_dim=tup(ec["encoder_dim"]),
                encoder_unmasked_dim=tup(ec["encoder_unmasked_dim"]), num_heads=tup(ec["num_heads"]),
                query_head_dim=tup(ec["query_head_dim"]), pos_head_dim=tup(ec["pos_head_dim"]),
                cnn_module_kernel=tup(ec["cnn_module_kernel"]), pos_dim=ec["pos_dim"])


def _fbank_one(args):
    """Worker of fbank_reference_style (top level: picklable for the spawn context)."""
    from oracle import fbank as ofb
    pcm, scale, high = args
    return ofb.fbank(pcm * scale, 80, high_freq=high).shape[0]


def fbank_reference_style(pcm, seconds, scale=32768.0, high=-400.0, workers=4):
    """The frontend timed the way the reference runs it: one call per
    utterance inside the DataLoader's worker processes (config `num_workers: 4`,
    task_factory/ctc_task.py:100) -- here the oracle's numpy fbank in 4 spawned processes (spawn:
    this process has already initialised the GPU), pool start-up excluded.  -> audio-s/s."""
    import multiprocessing as mp
    jobs = [(p, scale, high) for p in pcm]
    with mp.get_context("spawn").Pool(workers) as pool:
        pool.map(_fbank_one, jobs[:workers])                     # warm the workers (imports)
        t0 = time.perf_counter()
        for _ in range(3):
            pool.map(_fbank_one, jobs, chunksize=1)
        dt = (time.perf_counter() - t0) / 3
    return len(jobs) * seconds / dt


def cpu_baseline(cfg, state_dict, seconds=10.0, batch=8, n_labels=50, vocab=500, steps=3,
                 warmup=1):
    """Times the ORACLE train step on the host cores: numpy fbank, torch-CPU zipformer fwd+bwd
    (oracle/zipformer.py), stateless predictor + joiner projections (oracle/heads.py), k2-style
    simple + pruned RNN-T losses (oracle/k2_rnnt.py, C mutual-information recursion).  Uses ONLY
    oracle/ code and plain torch on `state_dict` (reference parameter names, CPU fp32 tensors);
    no speech2text_amd module runs here.  A reported baseline, never the product path."""
    import torch
    from oracle import fbank as ofb
    from oracle import heads as H
    from oracle import k2_rnnt as K2
    from oracle import zipformer as Z

    nthreads = host_threads()
    torch.set_num_threads(nthreads)
    os.environ.setdefault("OMP_NUM_THREADS", str(nthreads))
    sd = {k: v.detach().to("cpu", torch.float32).clone().requires_grad_(v.dtype.is_floating_point)
          for k, v in state_dict.items()}
    enc_sd = {k[len("_encoder.encoder."):]: v for k, v in sd.items()
              if k.startswith("_encoder.encoder.")}
    zcfg = _zcfg(cfg["encoder"]["config"])
    ctx = cfg["predictor"]["config"]["context_size"]
    prune = cfg["joiner"]["prune_range"]
    ss, ps = cfg["loss"]["simple_loss_scale"], cfg["loss"]["pruned_loss_scale"]
    rng = np.random.default_rng(20241218)
    pcm = synth_pcm(rng, batch, seconds)
    lab = torch.from_numpy(rng.integers(1, vocab - 1, size=(batch, n_labels)))
    lab_len = torch.full((batch,), n_labels, dtype=torch.int64)
    pyrand = random.Random(1234)
    times, loss = [], None
    for it in range(steps + warmup):
        t0 = time.perf_counter()
        feats = np.stack([ofb.fbank(p * 32768.0, 80, high_freq=-400.0) for p in pcm])
        x = torch.from_numpy(feats)
        lens = torch.full((batch,), feats.shape[1], dtype=torch.int64)
        ctl = Z.Ctl(training=True, rand=pyrand.random)
        y, ylen = Z.zipformer_forward(enc_sd, zcfg, x, lens, ctl, -1, -1)
        po = H.stateless_predictor(sd, "_predictor.predictor.", lab, ctx)
        am, lm = H.joiner_projections(sd, "_joiner.", y, po)
        logits, bnd, ranges, simple = K2.joiner_pruned(am, lm, lab, lab_len, ylen, prune)
        pruned = K2.rnnt_loss_pruned(logits, lab, ranges, 0, bnd)
        loss = H.pruned_rnnt_task_loss(simple, pruned, ss, ps)
        loss.backward()
        for v in sd.values():
            v.grad = None
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
    med = float(np.median(times))
    out = {"value": batch * seconds / med, "unit": "audio-seconds/sec", "cores": nthreads,
           "kind": "port",
           "sample": f"{batch} x {seconds:g}s utterances, {warmup} warm-up + {steps} timed steps "
                     f"(median {med:.2f} s) of the oracle C3 train step (numpy fbank + torch-CPU "
                     f"zipformer fwd/bwd + C/torch k2-style losses; parity unpinned for the k2 "
                     f"part), loss={float(loss.detach()):.4f}"}
    try:
        out["fbank_per_utterance_4_workers"] = {
            "value": fbank_reference_style(pcm, seconds), "unit": "audio-seconds/sec", "workers": 4,
            "sample": f"oracle numpy fbank, one call per utterance in 4 worker processes "
                      f"(reference DataLoader style), {batch} x {seconds:g}s, 3 passes"}
    except Exception as e:                                  # never fatal for the bench line
        out["fbank_per_utterance_4_workers"] = {"value": None, "sample": f"failed: {e}"}
    return out


def cpu_baseline_c2(cfg, state_dict, seconds=10.0, batch=4, n_labels=40, vocab=128, steps=3,
                    warmup=1):
    """Oracle conformer-CTC step (oracle/fbank.py, oracle/conformer.py, oracle/ctc.py)."""
    import torch
    from oracle import conformer as OC
    from oracle import fbank as ofb
    from oracle import heads as H

    nthreads = host_threads()
    torch.set_num_threads(nthreads)
    sd = {k: v.detach().to("cpu").clone() for k, v in state_dict.items()}
    for k, v in sd.items():                                 # buffers (BatchNorm running stats) stay plain
        if v.dtype.is_floating_point and "running_" not in k:
            v.requires_grad_(True)
    enc = {k[len("_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("_encoder.encoder.")}
    ec = cfg["encoder"]["config"]
    rng = np.random.default_rng(20241218)
    pcm = synth_pcm(rng, batch, seconds)
    lab = torch.from_numpy(rng.integers(1, vocab - 1, size=(batch, n_labels)))
    lab_len = torch.full((batch,), n_labels, dtype=torch.int64)
    times, loss = [], None
    for it in range(steps + warmup):
        t0 = time.perf_counter()
        feats = np.stack([ofb.fbank(p, 80) for p in pcm])
        x = torch.from_numpy(feats)
        lens = torch.full((batch,), feats.shape[1], dtype=torch.int64)
        y, ylen = OC.conformer_forward(enc, x, lens, ec["num_layers"], ec["num_heads"], training=True,
                                       dropout=ec.get("dropout", 0.0), seeds="torch")
        logits = H.projector(sd, "_decoder.decoder.", y)
        lp = logits.log_softmax(-1).transpose(0, 1)
        loss = torch.nn.functional.ctc_loss(lp, lab, ylen, lab_len, blank=0, reduction="mean",
                                            zero_infinity=True)
        loss.backward()
        for v in sd.values():
            v.grad = None
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
    med = float(np.median(times))
    return {"value": batch * seconds / med, "unit": "audio-seconds/sec", "cores": nthreads,
            "kind": "port",
            "sample": f"{batch} x {seconds:g}s utterances, {warmup} warm-up + {steps} timed steps "
                      f"(median {med:.2f} s) of the oracle C2 conformer-CTC train step, "
                      f"loss={float(loss.detach()):.4f}"}


def _cpu_time(step, batch, seconds, steps, warmup, what):
    times, loss = [], None
    for it in range(steps + warmup):
        t0 = time.perf_counter()
        loss = step()
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
    med = float(np.median(times))
    return {"value": batch * seconds / med, "unit": "audio-seconds/sec", "cores": host_threads(),
            "kind": "port",
            "sample": f"{batch} x {seconds:g}s utterances, {warmup} warm-up + {steps} timed steps "
                      f"(median {med:.2f} s) of the oracle {what} train step, loss={loss:.4f}"}


def _oracle_sd(state_dict):
    import torch
    sd = {k: v.detach().to("cpu").clone() for k, v in state_dict.items()}
    for k, v in sd.items():                                 # buffers (BatchNorm running stats) stay plain
        if v.dtype.is_floating_point and "running_" not in k and "_ssl_layer" not in k:
            v.requires_grad_(True)
    return sd


def cpu_baseline_c4(cfg, state_dict, seconds=10.0, batch=2, n_labels=60, vocab=128, steps=3,
                    warmup=1):
    """Oracle CTC_Hybrid_Rnnt step: numpy fbank, oracle conformer, LSTM predictor, unpruned joiner
    with out-projection (oracle/heads.py), full-lattice RNN-T loss (oracle/k2_rnnt.py) + CTC."""
    import torch
    from oracle import conformer as OC
    from oracle import fbank as ofb
    from oracle import heads as H
    from oracle import k2_rnnt as K2
    torch.set_num_threads(host_threads())
    sd = _oracle_sd(state_dict)
    enc = {k[len("_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("_encoder.encoder.")}
    ec, pc = cfg["encoder"]["config"], cfg["predictor"]["config"]
    rng = np.random.default_rng(20241218)
    pcm = synth_pcm(rng, batch, seconds)
    lab = torch.from_numpy(rng.integers(1, vocab - 1, size=(batch, n_labels)))
    lab_len = torch.full((batch,), n_labels, dtype=torch.int64)

    def step():
        feats = np.stack([ofb.fbank(p, 80) for p in pcm])
        x = torch.from_numpy(feats)
        lens = torch.full((batch,), feats.shape[1], dtype=torch.int64)
        y, ylen = OC.conformer_forward(enc, x, lens, ec["num_layers"], ec["num_heads"], training=True,
                                       dropout=ec.get("dropout", 0.0), seeds="torch")
        logits = H.projector(sd, "_decoder.decoder.", y)
        l_ctc = torch.nn.functional.ctc_loss(logits.log_softmax(-1).transpose(0, 1), lab, ylen,
                                             lab_len, blank=0, reduction="mean", zero_infinity=True)
        po = H.lstm_predictor(sd, "_predictor.predictor._predictor.", lab, pc["num_lstm_layers"],
                              pc["lstm_layer_norm"], pc["lstm_layer_norm_epsilon"])
        joint = H.joiner_full(sd, "_joiner.", y, po, cfg["joiner"]["activation"])
        l_rnnt = K2.rnnt_loss_full(joint, lab, ylen, lab_len)
        loss = H.hybrid_task_loss(l_rnnt, l_ctc, cfg["loss"]["rnnt_weight"], cfg["loss"]["ctc_weight"])
        loss.backward()
        for v in sd.values():
            v.grad = None
        return float(loss.detach())

    return _cpu_time(step, batch, seconds, steps, warmup, "C4 hybrid CTC + RNN-T")


def cpu_baseline_c5(cfg, state_dict, seconds=30.0, batch=2, steps=3, warmup=1):
    """Oracle BEST-RQ step: numpy fbank, quantizer labels (oracle/best_rq.py, fp64), half of the
    label frames masked with N(0, 0.1) noise, oracle conformer, Projector, masked KL."""
    import torch
    from oracle import best_rq as OB
    from oracle import conformer as OC
    from oracle import fbank as ofb
    from oracle import heads as H
    torch.set_num_threads(host_threads())
    sd = _oracle_sd(state_dict)
    enc = {k[len("_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("_encoder.encoder.")}
    ec = cfg["encoder"]["config"]
    pk = [k for k in sd if k.startswith("_ssl_layer") and "project" in k.lower()][0]
    cbk = [k for k in sd if k.startswith("_ssl_layer") and "codebook" in k.lower()][0]
    K = cfg["ssl_layer"]["layer_config"]["codebook_size"]
    rng = np.random.default_rng(20241218)
    pcm = synth_pcm(rng, batch, seconds)

    def step():
        feats = np.stack([ofb.fbank(p, 80) for p in pcm])
        lens = np.full((batch,), feats.shape[1], dtype=np.int64)
        labels = OB.make_labels(feats.astype(np.float64), sd[pk].numpy(), sd[cbk].numpy().reshape(1, K, 16))
        nl = OB.label_lengths(lens)
        T2 = labels.shape[2]
        mask = np.zeros((batch, T2), np.float32)
        x = feats.copy()
        for b in range(batch):
            idx = rng.choice(nl[b], nl[b] // 2, replace=False)
            mask[b, idx] = 1.0
            for t2 in idx:
                x[b, 4 * t2:4 * t2 + 7] = rng.normal(0.0, 0.1, size=(min(7, x.shape[1] - 4 * t2), 80))
        y, ylen = OC.conformer_forward(enc, torch.from_numpy(x), torch.from_numpy(lens),
                                       ec["num_layers"], ec["num_heads"], training=True,
                                       dropout=ec.get("dropout", 0.0), seeds="torch")
        logits = H.projector(sd, "_logits_layer.decoder.", y)
        loss = H.masked_kl_div(logits, torch.from_numpy(labels[0][:, :logits.shape[1]]),
                               torch.from_numpy(mask[:, :logits.shape[1]]), K + 1, 1.0, 0.1)
        loss.backward()
        for v in sd.values():
            v.grad = None
        return float(loss.detach())

    return _cpu_time(step, batch, seconds, steps, warmup, "C5 BEST-RQ SSL")


# ------------------------------------------------------------------ roofline bookkeeping
# fp32 products evaluated as six bf16 MFMA products (exact three-way split): the ceiling of THAT
# pipe is the dense bf16 peak / 6 -- reported next to the f32-MFMA fraction for the entries that
# run on it, so that a kernel on the bf16 cores is not flattered by the smaller f32 peak
BF16X3_CEILING_TFLOPS = 2500.0 / 6.0
# ... and as THREE products of the two-piece split (S2T_GEMM_ARITH=2, "bf16x2/3"): dense bf16 peak / 3
BF16X2_CEILING_TFLOPS = 2500.0 / 3.0


def gemm_arith():
    """(policy, ceiling TFLOP/s fp32-equivalent) of the arithmetic the library's bf16 GEMMs run in:
    'bf16x3/6' | 'bf16x2/3' when the forward, data-gradient and weight-gradient classes of product
    (include/s2t_mi355.h) run the same, else the three listed; the statistics class (Whiten's
    covariance and penalty products: six products by default) is in config.gemm_arith_classes.  The ceiling is that of the
    forward / data-gradient classes (what s2t_gemm_x3p serves); when those two differ, the HIGHER one
    (fewer products), so that the reported fraction is never flattered."""
    from speech2text_amd import zip_kernels as zk
    fd = min(zk.gemm_arith(zk.CLS_F), zk.gemm_arith(zk.CLS_D))
    return zk.gemm_arith_policy()[0], (BF16X3_CEILING_TFLOPS if fd == 3 else BF16X2_CEILING_TFLOPS)


def gemm_arith_classes():
    """{'F': .., 'D': .., 'W': .., 'S': ..}: the arithmetic of each class of product (forward, data
    gradient, weight gradient, statistics)."""
    from speech2text_amd import zip_kernels as zk
    return zk.gemm_arith_policy()[1]
BF16X3_ENTRIES = ("s2t_gemm_x3p", "s2t_gemm_x3p_bal", "s2t_gemm_x3p_map", "s2t_gemm_x3p_sq", "s2t_gemm_tn_grouped",
                  "s2t_gemm_f32", "s2t_gemm_f32_sq", "s2t_gemm_xtx", "s2t_conv3x3_gemm", "s2t_gemm_f32_batched")
# the class of product an entry (mostly) serves -- its ceiling is that class's arithmetic (the x3p entries
# serve forward AND data-gradient products: the higher ceiling of the two, gemm_arith())
ENTRY_CLASS = {"s2t_gemm_xtx": "S", "s2t_gemm_f32_sq": "S", "s2t_gemm_tn_grouped": "W", "s2t_gemm_f32": "W",
               "s2t_conv3x3_gemm": "W", "s2t_gemm_x3p_sq": "D"}


def _bound(p):
    """mfma when the call sites' algorithmic intensity exceeds the machine balance, else hbm."""
    balance = MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    return "mfma" if p["algo_flops"] > 0 and p["algo_flops"] > balance * p["algo_bytes"] else "hbm"


def roofline_report(prof_timed, prof_all, want, step_flops, ms_per_step, sampled_every=1):
    """`prof_timed`: HIP-event timings of the chosen entry point taken INSIDE the timed region.
    `prof_all`: every hand-written entry point, timed over extra (untimed) steps."""
    table = []
    arith_name, ceiling = gemm_arith()
    classes = gemm_arith_classes()

    def ceiling_of(entry):
        c = ENTRY_CLASS.get(entry)
        if c is None:
            return ceiling
        return BF16X3_CEILING_TFLOPS if classes[c] == "bf16x3/6" else BF16X2_CEILING_TFLOPS

    for name, p in sorted(prof_all.items(), key=lambda kv: -kv[1]["total_ms"]):
        row = {"entry": name, "launches_per_step": p["launches_per_step"],
               "ms_per_step": p["ms_per_step"], "avg_us": 1000.0 * p["avg_ms"]}
        if p["algo_bytes"] > 0 and p["total_ms"] > 0:
            gbs = p["algo_bytes"] / (p["total_ms"] * 1e-3) / 1e9
            row.update(bound=_bound(p), alg_GBps=gbs, frac_hbm=gbs / HBM_PEAK_GBS)
            if p["algo_flops"] > 0:
                tf = p["algo_flops"] / (p["total_ms"] * 1e-3) / 1e12
                row.update(alg_TFLOPs=tf, frac_mfma=tf / MFMA_F32_PEAK_TFLOPS)
                if name in BF16X3_ENTRIES:
                    row.update(frac_bf16_ceiling=tf / ceiling_of(name))
        table.append(row)
    out = {"bound": "hbm", "kernel": want, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": None, "traffic": None}
    p = prof_timed.get(want) if prof_timed else None
    if p and p["launches"] and p["algo_bytes"] > 0:
        if _bound(p) == "mfma":
            ach = p["algo_flops"] / (p["total_ms"] * 1e-3) / 1e12
            out.update(bound="mfma", achieved=ach, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                       frac=ach / MFMA_F32_PEAK_TFLOPS,
                       algorithmic_flops_per_launch=p["algo_flops"] / p["launches"])
            if want in BF16X3_ENTRIES:
                # the pipe this entry runs on: dense bf16 peak / products per fp32 term of the arithmetic
                out.update(gemm_arith=arith_name, frac_bf16_ceiling=ach / ceiling, bf16_ceiling_tflops=ceiling)
        else:
            ach = p["algo_bytes"] / (p["total_ms"] * 1e-3) / 1e9
            out.update(achieved=ach, frac=ach / HBM_PEAK_GBS)
        out.update(launches=p["launches"], sampled_every=sampled_every,
                   avg_launch_ms=p["avg_ms"],
                   algorithmic_bytes_per_launch=p["algo_bytes"] / p["launches"])
        if p.get("algo_bytes_min"):
            # operands every such product must move (A, C, the weight pieces) WITHOUT the epilogue's
            # extra operands and second outputs, which algorithmic_bytes_per_launch includes
            out["algorithmic_bytes_min_per_launch"] = p["algo_bytes_min"] / p["launches"]
    else:
        out["note"] = f"{want} was not launched in the timed region"
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath):
        try:
            ent = json.load(open(tpath)).get(want, {})
            out["traffic"] = ent.get("hbm_bytes_per_launch")
            # the algorithmic bytes of the SAME launches the counters were read over (the PMC passes
            # run bench.py --steps 1 --sample-every 1: their timed step is the counted step)
            if ent.get("algorithmic_bytes_per_launch"):
                out["traffic_algorithmic_bytes_per_launch"] = ent["algorithmic_bytes_per_launch"]
                out["traffic_over_algorithmic"] = out["traffic"] / ent["algorithmic_bytes_per_launch"]
        except Exception:
            pass
    out["kernels"] = table
    if step_flops:
        tf = step_flops / (ms_per_step * 1e-3) / 1e12
        out["step_mfma"] = {"flops_per_step": step_flops, "achieved_tflops": tf,
                            "peak_tflops": MFMA_F32_PEAK_TFLOPS, "frac": tf / MFMA_F32_PEAK_TFLOPS}
    return out


def gemm_paths():
    """Which code served the forward / data-gradient products of this process so far: our
    pre-split-weight kernel, the plan cache's choice (hipBLASLt or the round-3 kernel), and the
    ATen fallback (a shape hipBLASLt had no algorithm for) -- the last must stay 0 on the bench."""
    from speech2text_amd import zip_kernels as zk
    from speech2text_amd import _native as N
    # (counted inside the library: the native layer executor's launches never pass through Python)
    return {"x3p_calls": int(N.lib().s2t_gemm_x3p_calls()), "lt_calls": int(N.lib().s2t_linear_lt_calls()),
            "lt_own_calls": int(zk.lt_own_calls()), "aten_fallbacks": int(zk.LT_STATS["aten_fallbacks"])}


# ------------------------------------------------------------------ launcher
def self_launch(args, argv):
    """--gpus N>1 outside a launcher: start N ranks in child processes (this parent never
    touches the GPU) and relay the result."""
    port = int(os.environ.get("MASTER_PORT", 29500 + (os.getpid() % 2000)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def launcher_selftest(args, rank, world):
    """CPU-only rehearsal of the multi-rank protocol (gloo): rendezvous, barrier-bracketed timed
    loop, MAX over ranks, rank 0 prints the JSON line.  No GPU, no model: used by tests/."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.ones(1024)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1:
            dist.all_reduce(x)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "launcher-selftest", "value": args.steps / float(t.item()),
                          "unit": "steps/sec", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "backend": "gloo" if world > 1 else "none"}))
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------ main
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=["C3", "C2", "C4", "C5"])
    ap.add_argument("--batch", type=int, default=None, help="per-rank batch (utterances)")
    ap.add_argument("--seconds", type=float, default=None)
    ap.add_argument("--labels", type=int, default=None)
    ap.add_argument("--vocab", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--roofline-kernel", default=DEFAULT_ROOFLINE_KERNEL)
    ap.add_argument("--profile-steps", type=int, default=2,
                    help="extra untimed steps with every entry point bracketed by HIP events")
    ap.add_argument("--accum", type=int, default=1,
                    help="accumulate_grad_batches (the YAML trains with 20, "
                         "config/training/zipformer_stateless_pruned_rnnt.yaml:124): a step is "
                         "still ONE micro-batch pass (fbank+fwd+bwd); the gradient exchange, clip "
                         "and optimizer run on every accum-th step; --steps should be a multiple")
    ap.add_argument("--ddp-force", choices=["allreduce", "rs_ag"], default=None,
                    help="N = 1 only: run the data-parallel reducer on a 1-rank RCCL group (hooks, "
                         "buckets, exchange stream, collectives) to measure its overhead on one GPU")
    ap.add_argument("--random-chunk", action="store_true",
                    help="C3 only: the shipped YAML's chunk_size [16, 32, 64, -1] / "
                         "left_context_frames [64, 128, 256, -1] (one draw per step) instead of -1")
    ap.add_argument("--sample-every", type=int, default=0,
                    help="bracket every n-th launch of the roofline entry inside the timed region "
                         "(0 = chosen from its launches per step; 1 = every launch: the PMC passes of "
                         "tools/gpu_pmc2.sh use it so that traffic and algorithmic bytes cover the "
                         "same launches)")
    ap.add_argument("--launcher-selftest", action="store_true")
    args = ap.parse_args(argv)
    # per-rank batch, labels per utterance, vocabulary, utterance seconds
    d = {"C3": (64, 50, 500, 10.0), "C2": (32, 40, 128, 10.0), "C4": (16, 60, 128, 10.0),
         "C5": (8, 1, 128, 30.0)}[args.config]
    args.batch = args.batch or d[0]
    args.labels = args.labels or d[1]
    args.vocab = args.vocab or d[2]
    args.seconds = args.seconds or d[3]
    return args


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args, argv)                           # never returns

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.launcher_selftest:
        return launcher_selftest(args, rank, world)

    import torch
    import torch.distributed as dist

    def note(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    if world != args.gpus:
        note(f"--gpus {args.gpus} but WORLD_SIZE={world}: reporting n_gpus={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    elif args.ddp_force:
        os.environ["S2T_DDP_FORCE"] = "1"
        os.environ["S2T_DDP_ALGO"] = args.ddp_force
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)

    from speech2text_amd import _native
    from speech2text_amd.build_task import TaskFactory
    from speech2text_amd.trainer import Trainer

    c3 = args.config == "C3"
    cfg = {"C3": lambda: c3_config(args.vocab), "C2": lambda: c2_config(args.vocab),
           "C4": lambda: c4_config(args.vocab), "C5": c5_config}[args.config]()
    if args.random_chunk:
        if args.config != "C3":
            raise SystemExit("--random-chunk applies to the zipformer config (C3)")
        # config/training/zipformer_stateless_pruned_rnnt.yaml:65-66
        cfg["encoder"]["config"].update({"chunk_size": [16, 32, 64, -1],
                                         "left_context_frames": [64, 128, 256, -1]})
    if args.accum > 1:
        cfg["trainer"]["accumulate_grad_batches"] = int(args.accum)
    # reference build_task.py:47-48: EVERY rank seeds torch and `random` with 1234 -- all ranks then
    # draw the same chunk_size / Balancer / Whiten decisions per step (equal-cost steps: no rank
    # waits for another's slower launch list); only the DATA differs per rank (DistributedSampler
    # there, make_batch(rank) here)
    random.seed(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)                                 # same init on every rank
    task = TaskFactory.get(cfg["task"]["type"])(cfg)        # parameters are created on the host
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("timing the CPU baseline (oracle) ...")
        try:
            if args.config == "C3":
                cpu = cpu_baseline(cfg, task.state_dict(), args.seconds, args.cpu_batch,
                                   args.labels, args.vocab, steps=args.cpu_steps)
            elif args.config == "C2":
                cpu = cpu_baseline_c2(cfg, task.state_dict(), args.seconds, 4, args.labels,
                                      args.vocab, steps=args.cpu_steps)
            elif args.config == "C4":
                cpu = cpu_baseline_c4(cfg, task.state_dict(), args.seconds, 2, args.labels,
                                      args.vocab, steps=args.cpu_steps)
            else:
                cpu = cpu_baseline_c5(cfg, task.state_dict(), args.seconds, 2, steps=args.cpu_steps)
            note(f"cpu baseline: {cpu['value']:.2f} audio-s/s on {cpu['cores']} threads")
        except Exception as e:                              # a baseline failure must not abort
            note(f"cpu baseline FAILED: {type(e).__name__}: {e}")
            cpu = {"value": None, "unit": "audio-seconds/sec", "cores": host_threads(),
                   "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
    trainer = Trainer(**cfg["trainer"]).setup(task, device)
    task.train()
    torch.manual_seed(1234 + rank)
    batch = make_batch(rank, args.batch, args.seconds, args.labels, args.vocab, device)
    torch.manual_seed(1234)                                 # the training stream: as the reference, rank-independent

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    note("model + synthetic batch ready; warmup ...")
    loss = None
    want = args.roofline_kernel
    want_launches = 0
    for i in range(args.warmup):
        tw = time.perf_counter()
        last = i == args.warmup - 1
        if last and want == "auto":
            _native.profile_begin("*")
        loss = trainer.training_step(batch, i)
        torch.cuda.synchronize()
        if last and want == "auto":
            pw = _native.profile_end()
            cand = {k: v for k, v in pw.items() if v["algo_bytes"] > 0}
            want = max(cand, key=lambda k: cand[k]["total_ms"]) if cand else "s2t_relpos_attn_fwd"
            want_launches = cand[want]["launches"] if cand else 0
        note(f"warmup step {i}: {time.perf_counter() - tw:.3f} s, loss {float(loss):.4f}")
    if want == "auto":
        want = "s2t_relpos_attn_fwd"
    if world > 1:                                           # every rank brackets the SAME entry
        names = sorted(_native.parse_header())
        t = torch.tensor([names.index(want)], device=device, dtype=torch.int64)
        dist.broadcast(t, src=0)
        want = names[int(t.item())]
    # an entry launched hundreds of times per step is SAMPLED inside the timed region (every n-th
    # launch, n coprime to the launches per step): two event records per launch would cost the
    # timed step several per cent
    every = 1
    if args.sample_every > 0:
        every = args.sample_every
    elif want_launches > 48:
        every = max(2, want_launches // 32)
        while want_launches % every == 0 or (every > 2 and every % 2 == 0):
            every += 1
    sync()
    _native.profile_begin(want, every)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = trainer.training_step(batch, args.warmup + i)
    sync()
    dt = time.perf_counter() - t0
    prof_timed = _native.profile_end()
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss)
    prof_all = {}
    if args.profile_steps > 0:                              # every rank: the step has collectives
        _native.profile_begin("*")
        for i in range(args.profile_steps):
            trainer.training_step(batch, args.warmup + args.steps + i)
        torch.cuda.synchronize()
        prof_all = _native.profile_end()
        for p in prof_all.values():
            p["launches_per_step"] = p["launches"] / args.profile_steps
            p["ms_per_step"] = p["total_ms"] / args.profile_steps
    if world > 1:
        dist.barrier()
    audio_s = args.batch * args.seconds * world * args.steps
    ms = 1000.0 * dt / args.steps
    if rank == 0:
        # zipformer fwd+bwd ~3.6 GFLOP per audio-second; conformer ~6.7
        step_flops = (3.6e9 if c3 else 6.7e9) * args.batch * args.seconds
        name = {"C3": "zipformer pruned-RNN-T", "C2": "conformer-CTC",
                "C4": "conformer CTC_Hybrid_Rnnt", "C5": "conformer BEST-RQ SSL"}[args.config]
        workload = {
            "C3": "C3 zipformer pruned-RNN-T train step: chunk_size -1, 500 BPE, prune_range 5 "
                  "(fbank+fwd+bwd+allreduce+ScaledAdam)",
            "C2": "C2 conformer-CTC train step (fbank+fwd+bwd+allreduce+AdamW), 12 layers d=256, V=128",
            "C4": "C4 CTC_Hybrid_Rnnt train step (fbank+conformer 12x256+CTC head+LSTM predictor+"
                  "unpruned joiner w/ out-projection+RNN-T lattice loss, 0.8/0.2), V=128",
            "C5": "C5 BEST-RQ SSL train step (fbank+quantizer 8192x16+masked conformer 12x256+"
                  "Projector 8193+masked KL), 30 s clips"}[args.config]
        out = {
            "metric": f"audio-seconds/sec (train step, {name})",
            "value": audio_s / dt, "unit": "audio-seconds/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (workload.replace("chunk_size -1", "chunk_size 16/32/64/-1 random per step")
                                    if args.random_chunk else workload),
                       "accumulate_grad_batches": args.accum,
                       "ddp_forced_on_one_rank": args.ddp_force,
                       "gemm_paths": gemm_paths(),
                       "gemm_arith": gemm_arith()[0],
                       "gemm_arith_classes": gemm_arith_classes(),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "utterance_seconds": args.seconds, "labels_per_utt": args.labels,
                       "parallelism": f"dp{world}", "final_loss": final_loss},
            "roofline": roofline_report(prof_timed, prof_all, want, step_flops, ms, every),
            "cpu_baseline": cpu if cpu is not None else {
                "value": None, "unit": "audio-seconds/sec", "cores": host_threads(), "kind": "port",
                "sample": "not timed in this run (multi-rank or --no-cpu-baseline): the oracle "
                          "step is timed on rank 0 of the N=1 run, see that line"},
        }
        print(json.dumps(out), flush=True)
    if world > 1 or args.ddp_force:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
