#!/usr/bin/env python
"""Headline benchmark: audio-seconds/sec of one zipformer pruned-RNN-T training step.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (BASELINE.json metric, SURVEY.md section 8d "C3"): per-rank batch of 64 synthetic 16 kHz
10 s utterances (PCM resident in HBM before the timed region), 50 labels each from a 500-piece
vocabulary, zipformer YAML dims (config/training/zipformer_stateless_pruned_rnnt.yaml:49-95),
stateless predictor, pruned RNN-T (prune_range 5) with 0.5*simple + 0.5*pruned loss, fp32.
A step = on-GPU fbank -> CMVN -> Zipformer2 fwd -> predictor -> joiner (simple loss, prune
ranges, fused pruned lattice) -> backward -> bucketed RCCL gradient all-reduce (N>1) ->
grad-norm clip 5.0 -> ScaledAdam step -> Eden step.  Weak scaling: per-GPU work is fixed.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline / cpu_baseline
objects).
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR = 16000


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


def cpu_baseline(cfg, seconds=10.0, batch=2, n_labels=50, vocab=500, steps=2):
    """Times the oracle (our CPU restatement of the reference path: numpy fbank, torch-CPU
    zipformer fwd+bwd, k2-style RNN-T losses) on the host cores.  Test infrastructure used as
    a reported baseline only; never on the product path."""
    from oracle import fbank as ofb
    from oracle import k2_rnnt as K2
    from oracle import zipformer as Z
    from speech2text_amd.model.encoder.zipformer import Zipformer2, Zipformer2Config
    from speech2text_amd.model.joiner.joiner import JoinerConfig, Joiner
    from speech2text_amd.model.predictor.predictor import Predictor

    torch.manual_seed(1234)
    nthreads = host_threads()
    torch.set_num_threads(nthreads)
    ec = cfg["encoder"]["config"]
    enc = Zipformer2(Zipformer2Config(**ec))                # parameter container only
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    pred = Predictor(cfg["predictor"])
    join = Joiner(JoinerConfig(**cfg["joiner"]))
    ns = len(ec["downsampling_factor"])
    tup = lambda v: tuple(v) if isinstance(v, (list, tuple)) else (v,) * ns   # noqa: E731
    zcfg = dict(downsampling_factor=tup(ec["downsampling_factor"]),
                num_encoder_layers=tup(ec["num_encoder_layers"]), encoder_dim=tup(ec["encoder_dim"]),
                encoder_unmasked_dim=tup(ec["encoder_unmasked_dim"]), num_heads=tup(ec["num_heads"]),
                query_head_dim=tup(ec["query_head_dim"]), pos_head_dim=tup(ec["pos_head_dim"]),
                cnn_module_kernel=tup(ec["cnn_module_kernel"]), pos_dim=ec["pos_dim"])
    rng = np.random.default_rng(20241218)
    pcm = synth_pcm(rng, batch, seconds)
    lab = torch.from_numpy(rng.integers(1, vocab - 1, size=(batch, n_labels)))
    lab_len = torch.full((batch,), n_labels, dtype=torch.int64)
    pyrand = random.Random(1234)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        feats = np.stack([ofb.fbank(p * 32768.0, 80, high_freq=-400.0) for p in pcm])
        x = torch.from_numpy(feats)
        lens = torch.full((batch,), feats.shape[1], dtype=torch.int64)
        ctl = Z.Ctl(training=True, rand=pyrand.random)
        y, ylen = Z.zipformer_forward(sd, zcfg, x, lens, ctl, -1, -1)
        po, pl, _ = pred(lab, lab_len, pred.init_state())
        am = join._enc_proj(y)
        lm = join._pre_proj(po)
        logits, bnd, ranges, simple = K2.joiner_pruned(am, lm, lab, lab_len, ylen, 5)
        pruned = K2.rnnt_loss_pruned(logits, lab, ranges, 0, bnd)
        loss = 0.5 * simple + 0.5 * pruned
        loss.backward()
        for v in sd.values():
            v.grad = None
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
    med = float(np.median(times))
    return {"value": batch * seconds / med, "unit": "audio-seconds/sec", "cores": nthreads,
            "kind": "port",
            "sample": f"{batch} x {seconds:g}s utterances, {steps} timed steps (median) of the "
                      f"oracle train step (numpy fbank + torch-CPU zipformer fwd/bwd + k2-style "
                      f"losses), loss={float(loss.detach()):.4f}"}


# ------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-rank batch (utterances)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--labels", type=int, default=50)
    ap.add_argument("--vocab", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--roofline-kernel", default="relpos_attn_weights_fwd")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from speech2text_amd import _native
    from speech2text_amd.task_factory.rnnt_task import PrunedRnntTask
    from speech2text_amd.trainer import Trainer

    def note(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    cfg = c3_config(args.vocab)
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        note("timing the CPU baseline (oracle) ...")
        cpu = cpu_baseline(cfg, args.seconds, args.cpu_batch, args.labels, args.vocab)
        note(f"cpu baseline: {cpu['value']:.2f} audio-s/s on {cpu['cores']} threads")

    random.seed(1234 + rank)
    np.random.seed(1234 + rank)
    torch.manual_seed(1234)                                 # same init on every rank
    task = PrunedRnntTask(cfg)
    trainer = Trainer(**cfg["trainer"]).setup(task, device)
    task.train()
    torch.manual_seed(1234 + rank)
    batch = make_batch(rank, args.batch, args.seconds, args.labels, args.vocab, device)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    note("model + synthetic batch ready; warmup ...")
    loss = None
    for i in range(args.warmup):
        tw = time.perf_counter()
        loss = trainer.training_step(batch, i)
        torch.cuda.synchronize()
        note(f"warmup step {i}: {time.perf_counter() - tw:.3f} s, loss {float(loss):.4f}")
    sync()
    _native.profile_begin(args.roofline_kernel)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = trainer.training_step(batch, args.warmup + i)
    sync()
    dt = time.perf_counter() - t0
    prof = _native.profile_end()
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    audio_s = args.batch * args.seconds * world * args.steps
    if rank == 0:
        from speech2text_amd import roofline
        out = {
            "metric": "audio-seconds/sec (train step, zipformer pruned-RNN-T)",
            "value": audio_s / dt, "unit": "audio-seconds/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C3 zipformer-stateless pruned-RNN-T train step (fbank+fwd+bwd"
                                   "+allreduce+ScaledAdam), 500 BPE, prune_range 5, chunk_size -1",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "utterance_seconds": args.seconds, "labels_per_utt": args.labels,
                       "parallelism": f"dp{world}", "final_loss": float(loss)},
            "roofline": roofline.report(args.roofline_kernel, prof, args),
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
