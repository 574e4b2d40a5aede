"""Generate golden vectors under tests/golden/ by RUNNING the reference here.

Runs only in the build container (needs /root/reference).  Writes plain arrays
(inputs, parameters, expected outputs) -- never reference source.  Usage:
    python tools/gen_golden.py [fbank] [ctc] [bestrq] [zipformer] [losses] ...
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")


def synth_pcm(rng, n):
    t = np.arange(n) / 16000.0
    x = 0.05 * rng.standard_normal(n)
    for _ in range(3):
        f = rng.uniform(100, 4000)
        x += rng.uniform(0.02, 0.2) * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28))
    return np.clip(x, -1, 1).astype(np.float32)


def gen_fbank():
    import torch
    m = torch.jit.load("/root/reference/sample_data/model/frontend.script")
    rng = np.random.default_rng(20241218)
    out = {}
    for i, n in enumerate([400, 559, 560, 12560, 54080, 16000 * 3 + 77]):
        pcm = synth_pcm(rng, n)
        y = m(torch.from_numpy(pcm)[None]).numpy()
        out[f"pcm{i}"] = pcm
        out[f"feat{i}"] = y
    np.savez_compressed(os.path.join(OUT, "fbank_script64.npz"), **out)
    print("fbank:", {k: v.shape for k, v in out.items()})


def gen_ctc():
    import torch
    ref_import.install_stubs()
    from model.loss.ctc_loss import CtcLoss, CtcLossConfig
    rng = np.random.default_rng(7)
    out = {}
    cases = [(4, 30, 16, 8), (3, 50, 40, 12), (2, 12, 8, 10), (5, 64, 128, 20)]
    for ci, (B, T, V, U) in enumerate(cases):
        logits = torch.from_numpy(rng.standard_normal((B, T, V)).astype(np.float32) * 2.0)
        tl = torch.from_numpy(rng.integers(1, U + 1, size=B)).long()
        tl[0] = U
        il = torch.from_numpy(rng.integers(T // 2, T + 1, size=B)).long()
        il[0] = T
        if ci == 2:
            il[1] = 3  # infeasible: too few frames -> inf -> zero_infinity
            tl[1] = 10
        tg = torch.from_numpy(rng.integers(1, V, size=(B, U))).long()
        if ci == 1:
            tg[0, 1] = tg[0, 0]  # repeated label
            tg[0, 2] = tg[0, 0]
        for b in range(B):
            tg[b, tl[b]:] = 0
        logits.requires_grad_(True)
        loss = CtcLoss(CtcLossConfig())(logits, tg, il, tl)
        loss.backward()
        out[f"logits{ci}"] = logits.detach().numpy()
        out[f"targets{ci}"] = tg.numpy()
        out[f"in_len{ci}"] = il.numpy()
        out[f"tgt_len{ci}"] = tl.numpy()
        out[f"loss{ci}"] = loss.detach().numpy()
        out[f"grad{ci}"] = logits.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "ctc_ref.npz"), **out)
    print("ctc losses:", [float(out[f"loss{i}"]) for i in range(len(cases))])


def gen_bestrq():
    import torch
    ref_import.install_stubs()
    from model.ssl.best_rq import BestRQLayer, BestRQLayerConfig, MaskingStrategyConfig
    out = {}
    for ci, (basis, K, D, ncb, B, T) in enumerate([("cosine", 256, 16, 1, 3, 203),
                                                   ("euclidean", 64, 8, 2, 2, 131),
                                                   ("cosine", 8192, 16, 1, 2, 399)]):
        torch.manual_seed(1234 + ci)
        np.random.seed(99 + ci)
        layer = BestRQLayer(BestRQLayerConfig(feat_dim=80, num_codebooks=ncb, codebook_dim=D,
                                              codebook_size=K, label_basis=basis),
                            MaskingStrategyConfig(mask_proportion=0.5, mean_span_length=1,
                                                  span_select_type="static", seed=5))
        raw = torch.randn(B, T, 80) * 3.0
        aug = raw + 0.1 * torch.randn(B, T, 80)
        length = torch.tensor([T] + [T - 17 * (i + 1) for i in range(B - 1)])
        aug_in = aug.clone()
        res = layer(raw.clone(), aug_in, length)
        out[f"projector{ci}"] = layer._projector.detach().numpy()
        for j in range(ncb):
            out[f"codebook{ci}_{j}"] = layer._codebooks[j].detach().numpy()
        out[f"raw{ci}"] = raw.numpy()
        out[f"aug{ci}"] = aug.numpy()
        out[f"length{ci}"] = length.numpy()
        out[f"labels{ci}"] = res["labels"].numpy()
        out[f"masked_dim{ci}"] = res["masked_dim"].numpy()
        out[f"basis{ci}"] = np.array(basis)
        # masked positions of feats (values are random noise; only positions are pinned)
        out[f"changed{ci}"] = (res["masked_feats"] != aug).any(-1).numpy()
    np.savez_compressed(os.path.join(OUT, "bestrq_ref.npz"), **out)
    print("bestrq:", {k: v.shape for k, v in out.items() if k.startswith("labels")})




ZIP_TINY = dict(feature_dim=80, downsampling_factor=(1, 2, 4), num_encoder_layers=(1, 1, 1),
                feedforward_dim=(64, 96, 96), encoder_dim=(32, 48, 48),
                encoder_unmasked_dim=(24, 32, 32), num_heads=(4, 4, 4), query_head_dim=(8,),
                value_head_dim=(4,), pos_head_dim=(4,), pos_dim=16, cnn_module_kernel=(7, 5, 5),
                causal=True)


def _tiny_zipformer(chunk, left, **extra):
    import torch
    from model.encoder.zipformer import Zipformer2, Zipformer2Config
    torch.manual_seed(1234)
    cfg = Zipformer2Config(**ZIP_TINY, chunk_size=chunk, left_context_frames=left, **extra)
    m = Zipformer2(cfg)
    # move parameters away from their special initial values so every term matters
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bypass_scale"):
                p.uniform_(0.2, 0.9)
            elif n.endswith("chunkwise_conv_scale"):
                p.normal_(0, 0.3)
            elif n.endswith("downsample.bias") or n.endswith("downsample_output.bias"):
                p.normal_(0, 0.5)
            elif n.endswith("log_scale"):
                p.fill_(0.7)
            elif "out_proj" in n or "pointwise_conv2" in n or "linear_pos" in n:
                p.mul_(8.0)
    return m


def gen_zipformer_stream():
    """Tiny causal Zipformer2 (chunk 8, left context 16, CTC projection on): `streaming_step`
    over 6 consecutive chunks from `get_init_states`; per chunk the encoder output before the
    projection, the log-softmax output, and the final states.  Also the chunked non-streaming
    forward of the whole utterance for the streaming-vs-chunked property."""
    import torch
    ref_import.install_stubs()
    chunk, left, nchunks, B = 8, 16, 6, 2
    m = _tiny_zipformer((chunk,), (left,), for_ctc=True, num_tokens=13)
    m.eval()
    T = 2 * chunk + 13
    g = torch.Generator().manual_seed(321)
    feats = torch.randn(B, 2 * chunk * (nchunks - 1) + T, 80, generator=g) * 2.0
    out = {"feats": feats.numpy(), "chunk": np.int64(chunk), "left": np.int64(left)}
    for k, v in m.state_dict().items():
        out["sd." + k] = v.numpy()
    states = m.get_init_states(B)
    out["n_states"] = np.int64(len(states))
    for i, s in enumerate(states):
        out[f"init_shape.{i}"] = np.array(s.shape, dtype=np.int64)
    for c in range(nchunks):
        x = feats[:, 2 * chunk * c:2 * chunk * c + T]
        m._for_ctc = False
        raw, _ = m.streaming_step(x, [s.clone() for s in states])
        m._for_ctc = True
        y, states = m.streaming_step(x, states)
        out[f"raw.{c}"] = raw.numpy()
        out[f"out.{c}"] = y.numpy()
        if c in (0, 2):                       # warm-up (cache partly empty) and steady state
            for i, s in enumerate(states):
                out[f"state{c}.{i}"] = s.numpy()
    for i, s in enumerate(states):
        out[f"final_state.{i}"] = s.numpy()
    with torch.no_grad():
        lens = torch.full((B,), feats.shape[1], dtype=torch.int64)
        yf, yl = m(feats, lens)
    out["chunked_out"] = yf.numpy()
    out["chunked_lens"] = yl.numpy()
    np.savez_compressed(os.path.join(OUT, "zipformer_tiny_stream.npz"), **out)
    print("zipformer_stream:", out["raw.0"].shape, out["out.0"].shape, len(states), "states")


def gen_zipformer():
    """Tiny Zipformer2: eval forward, and a deterministic training step (random.random == 0:
    every Balancer / Whiten / limit_param_value / attention-score penalty fires) with grads."""
    import random
    import torch
    ref_import.install_stubs()
    from model.encoder.zipformer import Zipformer2, Zipformer2Config
    for tag, chunk, left in [("full", (-1,), (-1,)), ("chunk8", (8,), (16,))]:
        m = _tiny_zipformer(chunk, left)
        g = torch.Generator().manual_seed(99)
        x = torch.randn(3, 77, 80, generator=g) * 2.0
        lens = torch.tensor([77, 60, 41])
        out = {"x": x.numpy(), "lens": lens.numpy()}
        for k, v in m.state_dict().items():
            out["sd." + k] = v.numpy()
        m.eval()
        with torch.no_grad():
            y, yl = m(x, lens)
        out["eval_out"] = y.numpy()
        out["eval_lens"] = yl.numpy()
        # deterministic training step
        m.train()
        real_random = random.random
        random.random = lambda: 0.0
        try:
            torch.manual_seed(7)
            xt = x.clone().requires_grad_(True)
            y, yl = m(xt, lens)
            wts = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
            loss = (y * wts).sum()
            loss.backward()
        finally:
            random.random = real_random
        out["train_out"] = y.detach().numpy()
        out["train_wts"] = wts.numpy()
        out["train_loss"] = loss.detach().numpy()
        out["grad.x"] = xt.grad.numpy()
        for n, p in m.named_parameters():
            out["grad." + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        np.savez_compressed(os.path.join(OUT, f"zipformer_tiny_{tag}.npz"), **out)
        print("zipformer", tag, y.shape, float(loss))


def gen_scaledadam():
    """30 steps of the reference ScaledAdam (+ Eden) on 5 small tensors, seeded grads."""
    import torch
    ref_import.install_stubs()
    from optimizer.scaled_adam import ScaledAdam
    from optimizer.optim_setup import Eden
    torch.manual_seed(0)
    shapes = [(5, 7), (3,), (), (5, 7), (4, 2, 3)]
    ps = [torch.nn.Parameter(torch.randn(s) * (0.1 if i == 3 else 1.0)) for i, s in enumerate(shapes)]
    out = {f"init{i}": p.detach().numpy().copy() for i, p in enumerate(ps)}
    opt = ScaledAdam(ps, lr=0.045, clipping_scale=2.0, clipping_update_period=6)
    sched = Eden(opt, lr_batches=10, warmup_batches=4)
    g = torch.Generator().manual_seed(1)
    lrs = []
    for it in range(30):
        for i, p in enumerate(ps):
            gr = torch.randn(p.shape, generator=g) * (5.0 if it % 7 == 3 else 1.0)
            out[f"grad{it}_{i}"] = gr.numpy()
            p.grad = gr.clone()
        opt.step()
        sched.step()
        lrs.append(opt.param_groups[0]["lr"])
        if it in (0, 9, 29):
            for i, p in enumerate(ps):
                out[f"p{it}_{i}"] = p.detach().numpy().copy()
    out["lrs"] = np.array(lrs)
    np.savez_compressed(os.path.join(OUT, "scaledadam_ref.npz"), **out)
    print("scaledadam lrs", lrs[:3], lrs[-1])


def gen_scaledadam_state():
    """The reference ScaledAdam's state_dict after 15 of the 30 steps of gen_scaledadam (same
    seeds), flattened to arrays, plus the parameters at that point: the checkpoint-interchange
    fixture (optimizer/scaled_adam.py:30-109 keeps the state per batch of same-shaped tensors)."""
    import torch
    ref_import.install_stubs()
    from optimizer.scaled_adam import ScaledAdam
    from optimizer.optim_setup import Eden
    torch.manual_seed(0)
    shapes = [(5, 7), (3,), (), (5, 7), (4, 2, 3)]
    ps = [torch.nn.Parameter(torch.randn(s) * (0.1 if i == 3 else 1.0)) for i, s in enumerate(shapes)]
    opt = ScaledAdam(ps, lr=0.045, clipping_scale=2.0, clipping_update_period=6)
    sched = Eden(opt, lr_batches=10, warmup_batches=4)
    g = torch.Generator().manual_seed(1)
    for it in range(15):
        for i, p in enumerate(ps):
            p.grad = (torch.randn(p.shape, generator=g) * (5.0 if it % 7 == 3 else 1.0)).clone()
        opt.step()
        sched.step()
    sd = opt.state_dict()
    out = {f"p14_{i}": p.detach().numpy().copy() for i, p in enumerate(ps)}
    keys = []
    for idx, e in sd["state"].items():
        for k, v in e.items():
            out[f"state{idx}_{k}"] = v.numpy() if torch.is_tensor(v) else np.asarray(v)
            keys.append(f"{idx}:{k}")
    out["keys"] = np.array(keys)
    out["lr"] = np.asarray(sd["param_groups"][0]["lr"])
    np.savez_compressed(os.path.join(OUT, "scaledadam_state_ref.npz"), **out)
    print("scaledadam state keys", keys)


def gen_subsampling():
    """Reference conformer Subsampling (rates 4/6/8); torchaudio is stubbed (third-party)."""
    import sys, types, importlib.machinery
    import torch
    ref_import.install_stubs()
    if "torchaudio" not in sys.modules:
        ta = types.ModuleType("torchaudio")
        ta.__spec__ = importlib.machinery.ModuleSpec("torchaudio", None)
        ta.models = types.SimpleNamespace(Conformer=None)
        sys.modules["torchaudio"] = ta
    from model.encoder.conformer import Subsampling
    out = {}
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 120, 80, generator=g)
    lens = torch.tensor([120, 97, 50])
    out["x"], out["lens"] = x.numpy(), lens.numpy()
    for rate in (4, 6, 8):
        torch.manual_seed(rate)
        m = Subsampling(80, 32, rate)
        y, yl = m(x, lens)
        for k, v in m.state_dict().items():
            out[f"sd{rate}.{k}"] = v.numpy()
        out[f"out{rate}"] = y.detach().numpy()
        out[f"len{rate}"] = yl.numpy()
    np.savez_compressed(os.path.join(OUT, "subsampling_ref.npz"), **out)
    print("subsampling", {r: out[f"out{r}"].shape for r in (4, 6, 8)})


def gen_heads():
    """StatelessPredictor, MaskedKLDivergence, MaskedCELoss, GlobalCmvnLayer, Projector run from
    the reference tree; plus the task-level loss-combination formulas evaluated on the scalars
    the reference classes returned (task_factory/rnnt_task.py:349,496-499; ssl_task.py:140-167)."""
    import torch
    ref_import.install_stubs()
    from model.predictor.stateless_predictor import StatelessPredictor, StatelessPredictorConfig
    from model.loss.kl_divergence import MaskedKLDivergence, MaskedKLDivergenceConfig
    from model.loss.cross_entropy import MaskedCELoss, MaskedCELossConfig
    from model.layer.global_cmvn import GlobalCmvnLayer
    from model.decoder.projector import Projector, ProjectorConfig
    torch.manual_seed(20241218)
    rng = np.random.default_rng(20241218)
    out = {}
    # ---- stateless predictor (fwd + grads of all parameters)
    for ci, (V, D, E, ctx, B, U) in enumerate([(64, 48, 32, 5, 3, 7), (500, 32, 48, 5, 2, 12),
                                               (32, 16, 24, 2, 2, 5), (32, 16, 24, 1, 2, 4)]):
        m = StatelessPredictor(StatelessPredictorConfig(num_symbols=V, output_dim=D,
                                                        symbol_embedding_dim=E, context_size=ctx))
        lab = torch.from_numpy(rng.integers(1, V, size=(B, U))).long()
        y, ln, st = m(lab, torch.full((B,), U), m.init_state())
        w = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
        (y * w).sum().backward()
        out[f"pred{ci}_cfg"] = np.array([V, D, E, ctx])
        out[f"pred{ci}_labels"] = lab.numpy()
        out[f"pred{ci}_out"] = y.detach().numpy()
        out[f"pred{ci}_state"] = st.numpy()
        out[f"pred{ci}_w"] = w.numpy()
        for k, v in m.state_dict().items():
            out[f"pred{ci}_sd_{k}"] = v.numpy()
        for k, v in m.named_parameters():
            out[f"pred{ci}_grad_{k}"] = v.grad.numpy()
    # ---- masked KL / CE (2-D masks, 1-D length masks, smoothing, scale)
    for ci, (B, T, K, eps, scale) in enumerate([(3, 11, 17, 0.0, 1.0), (2, 9, 33, 0.1, 1.0),
                                                (2, 3, 8193, 0.1, 1.0), (2, 7, 13, 0.2, 0.5)]):
        lens = torch.tensor([T] + [int(x) for x in rng.integers(2, T, size=B - 1)])
        lab = torch.from_numpy(rng.integers(1, K, size=(B, T))).long()
        m2 = (torch.from_numpy(rng.random((B, T))) < 0.5) & (torch.arange(T)[None] < lens[:, None])
        m2[0, 0] = True
        m2 = m2.float()
        base = torch.from_numpy(rng.standard_normal((B, T, K)).astype(np.float32) * 2.0)
        out[f"ssl{ci}_cfg"] = np.array([K, eps, scale])
        out[f"ssl{ci}_logits"] = base.numpy()
        out[f"ssl{ci}_labels"] = lab.numpy()
        out[f"ssl{ci}_mask2d"] = m2.numpy()
        out[f"ssl{ci}_lens"] = lens.numpy()
        for name, cls, cfgcls in (("kl", MaskedKLDivergence, MaskedKLDivergenceConfig),
                                  ("ce", MaskedCELoss, MaskedCELossConfig)):
            mod = cls(cfgcls(num_classes=K, scale_factor=scale, label_smoothing=eps))
            for mname, mk in (("mask", m2), ("len", lens)):
                lg = base.clone().requires_grad_(True)
                loss = mod(lg * 1.0, lab, mk)     # "* 1.0": the class scales its input in place
                loss.backward()
                out[f"ssl{ci}_{name}_{mname}_loss"] = loss.detach().numpy()
                out[f"ssl{ci}_{name}_{mname}_grad"] = lg.grad.numpy()
        # the SSL task's combination over "codebooks" (ssl_task.py:140-167) on these scalars
        ml = [torch.tensor(float(out[f"ssl{ci}_kl_mask_loss"])), torch.tensor(1.25)]
        tl = [torch.tensor(float(out[f"ssl{ci}_kl_len_loss"])), torch.tensor(0.75)]
        out[f"ssl{ci}_task"] = np.array([float(sum(ml) / 2), float(sum(tl) / 2)])
    # ---- GlobalCmvn + Projector
    cm = GlobalCmvnLayer({"feat_type": "fbank", "feat_config": {"num_mel_bins": 80}})
    cm.global_mean.copy_(torch.from_numpy(rng.standard_normal(80).astype(np.float32)))
    cm.global_istd.copy_(torch.from_numpy(rng.uniform(0.2, 2.0, 80).astype(np.float32)))
    x = torch.from_numpy(rng.standard_normal((2, 13, 80)).astype(np.float32) * 5)
    out["cmvn_x"], out["cmvn_mean"], out["cmvn_istd"] = x.numpy(), cm.global_mean.numpy(), \
        cm.global_istd.numpy()
    out["cmvn_y"] = cm(x).numpy()
    pj = Projector(ProjectorConfig(input_dim=24, output_dim=40, dropout_p=0.0))
    xp = torch.from_numpy(rng.standard_normal((2, 9, 24)).astype(np.float32))
    yp, _ = pj(xp, torch.tensor([9, 5]))
    out["proj_x"], out["proj_y"] = xp.numpy(), yp.detach().numpy()
    for k, v in pj.state_dict().items():
        out[f"proj_sd_{k}"] = v.numpy()
    # ---- RNN-T task formulas (plain float arithmetic in the reference, rnnt_task.py:349,496-499)
    s, pr, c = 3.25, 1.5, 0.625
    out["formula_pruned"] = np.array([0.5 * s + 0.5 * pr, 0.5 * s + 0.5 * pr + c], np.float64)
    out["formula_hybrid"] = np.array([0.8 * s + 0.2 * c], np.float64)
    np.savez_compressed(os.path.join(OUT, "heads_ref.npz"), **out)
    print("heads:", len(out), "arrays")


def gen_aug():
    """SpecAugment / MixFeats / AddNoise / DynamicBucketBatchSampler run from the reference tree
    (third-party torchaudio / pytorch_lightning stubbed; they are not touched by these classes)
    with Python `random` seeded: inputs, the draws' seed and the outputs."""
    import importlib.machinery
    import random
    import types
    import torch
    ref_import.install_stubs()
    for name in ("torchaudio", "torchaudio.sox_effects", "sentencepiece"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = types.ModuleType(name)
                m.__spec__ = importlib.machinery.ModuleSpec(name, None)
                sys.modules[name] = m
    from dataset.frontend.data_augmentation import AddNoise, MixFeats, SpecAugment
    rng = np.random.default_rng(99)
    out = {}
    # ---- SpecAugment: 3 utterances, sequential calls under one seed
    feats = [torch.from_numpy(rng.standard_normal((T, 80)).astype(np.float32)) for T in (120, 57, 300)]
    random.seed(4321)
    sa = SpecAugment(num_t_mask=2, num_f_mask=2, max_t=50, max_f=10)
    for i, f in enumerate(feats):
        out[f"sa_in{i}"] = f.numpy()
        out[f"sa_out{i}"] = sa.process(f).numpy()
    # ---- MixFeats: noise shorter and longer than the source
    random.seed(4322)
    mf = MixFeats(snrs=(10, 20))
    for i, (T, Tn) in enumerate([(100, 37), (64, 200), (150, 150)]):
        src = torch.from_numpy((rng.standard_normal((T, 80)) * 2 - 3).astype(np.float32))
        nz = torch.from_numpy((rng.standard_normal((Tn, 80)) * 2 - 5).astype(np.float32))
        out[f"mf_src{i}"], out[f"mf_noise{i}"] = src.numpy(), nz.numpy()
        out[f"mf_out{i}"] = mf.process(src, nz).numpy()
    # ---- AddNoise (it scales its noise argument in place: pass a copy)
    random.seed(4323)
    an = AddNoise(min_snr_db=10, max_snr_db=50)
    for i, (n, m) in enumerate([(4000, 1500), (2000, 6000), (3001, 3001)]):
        pcm = torch.from_numpy((rng.standard_normal((1, n)) * 0.1).astype(np.float32))
        nz = torch.from_numpy((rng.standard_normal((1, m)) * 0.05).astype(np.float32))
        out[f"an_pcm{i}"], out[f"an_noise{i}"] = pcm.numpy(), nz.numpy()
        out[f"an_out{i}"] = an.process(pcm.clone(), nz.clone()).numpy()
    # ---- DynamicBucketBatchSampler on a synthetic duration table
    try:
        for _ in range(12):                                  # stub whatever third-party wheel is absent
            try:
                from dataset.sampler import DynamicBucketBatchSampler
                break
            except ModuleNotFoundError as e:
                name = e.name
                assert not os.path.exists(os.path.join("/root/reference", name.split(".")[0])), name
                m = types.ModuleType(name)
                m.__spec__ = importlib.machinery.ModuleSpec(name, None)
                m.__path__ = []
                m.__getattr__ = lambda attr, _n=name: type(attr, (), {})
                sys.modules[name] = m
        durs = rng.uniform(1.0, 15.0, size=400)

        class DS:
            lower_bound, high_bound, total_data_amount = 1.0, 15.0, float(durs.sum())

            def fetch_data_k_info(self, i, k="duration"):
                return float(durs[i])

        class SM:
            rank, num_replicas = 0, 1

            def __iter__(self):
                return iter(range(400))

            def __len__(self):
                return 400

        bs = DynamicBucketBatchSampler(SM(), DS(), num_bucket=6, min_batch_size=4, volume_threshold=60)
        it = iter(bs)
        batches = [next(it) for _ in range(25)]
        out["bs_durs"] = durs
        out["bs_len"] = np.array([len(bs)])
        out["bs_sizes"] = np.array([len(b) for b in batches])
        out["bs_flat"] = np.array([i for b in batches for i in b])
    except Exception as e:                                  # dataset.dataset needs more wheels
        print("sampler golden skipped:", type(e).__name__, e)
    np.savez_compressed(os.path.join(OUT, "aug_ref.npz"), **out)
    print("aug:", len(out), "arrays")


def gen_state_keys():
    """Checkpoint interchange fixture: parameter / buffer names and shapes of the reference
    modules the Pruned_Rnnt task holds, built from the shipped zipformer YAML
    (task_factory/rnnt_task.py:56-64,444-445 attribute names -> Lightning `state_dict` prefixes)."""
    import json, sys, types, importlib.machinery
    import torch, yaml
    ref_import.install_stubs()
    if "torchaudio" not in sys.modules:
        ta = types.ModuleType("torchaudio")
        ta.__spec__ = importlib.machinery.ModuleSpec("torchaudio", None)
        ta.models = types.SimpleNamespace(Conformer=None, Emformer=None)
        ta.functional = types.SimpleNamespace(rnnt_loss=None)
        sys.modules["torchaudio"] = ta
    cfg = yaml.safe_load(open("/root/reference/config/training/zipformer_stateless_pruned_rnnt.yaml"))
    from model.encoder.zipformer import Zipformer2, Zipformer2Config
    from model.predictor.stateless_predictor import StatelessPredictor, StatelessPredictorConfig
    from model.joiner.joiner import Joiner, JoinerConfig
    from model.decoder.decoder import Decoder
    mods = {"_encoder.encoder": Zipformer2(Zipformer2Config(**cfg["encoder"]["config"])),
            # model/predictor/predictor.py:27 holds it as `.predictor` (the factory module itself
            # imports the torchaudio-based LSTM predictor, a third-party wheel absent here)
            "_predictor.predictor": StatelessPredictor(StatelessPredictorConfig(**cfg["predictor"]["config"])),
            "_joiner": Joiner(config=JoinerConfig(**cfg["joiner"])),
            "_decoder": Decoder(cfg["decoder"])}
    if "ctc_projector" in cfg:
        mods["_ctc_projector"] = Decoder(cfg["ctc_projector"])
    out = {}
    for pre, m in mods.items():
        for k, v in m.state_dict().items():
            out[f"{pre}.{k}"] = list(v.shape)
    json.dump(out, open(os.path.join(OUT, "state_keys_c3.json"), "w"), indent=0, sort_keys=True)
    print("state_keys", len(out))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["fbank", "ctc", "bestrq", "zipformer", "scaledadam"]
    for w in which:
        globals()["gen_" + w]()
