"""GPU: which class of product (forward / data gradient / weight gradient / statistics) decides the
gradient error of the two-piece arithmetic?  The C3 YAML-dims training step of
tests/test_gpu_full_configs.py::test_c3_yaml_dims_training_step_gradients_vs_oracle, the oracle's
gradients computed ONCE per (rv, chunk) case, the product path run under every policy of POLICIES
(S2T_GEMM_ARITH_F/_D/_W/_S are read per call); prints the loss error and the worst parameters.
usage: python tools/exp_arith_policy.py [case ...]   (cases: indices into CASES)"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import heads as H  # noqa: E402
from oracle import k2_rnnt as K2  # noqa: E402
from oracle import zipformer as Z  # noqa: E402
from speech2text_amd import flat, rng, zip_layer, zip_native  # noqa: E402
from speech2text_amd.build_task import TaskFactory  # noqa: E402

CASES = [(0.0, (-1, -1)), (0.2, (-1, -1)), (0.2, (32, 128)), (0.0, (16, 64)), (0.2, (64, 256))]
POLICIES = [p for p in os.environ.get("POLICIES", "3333,2222,3222,2322,2232,2223,3322,3223,2233").split(",")]
dev = torch.device("cuda")
rng.rand = lambda *s, device=None, dtype=torch.float32: torch.rand(*s, dtype=dtype).to(device)


def set_policy(p):
    for k, v in zip("FDWS", p):
        os.environ["S2T_GEMM_ARITH_" + k] = v


def build(cfg):
    random.seed(1234)
    torch.manual_seed(1234)
    task = TaskFactory.get("Pruned_Rnnt")(cfg)
    pn = {n for n, _ in task.named_parameters()}
    sd = {k: v.detach().cpu().clone().requires_grad_(k in pn and v.dtype.is_floating_point)
          for k, v in task.state_dict().items()}
    task.to(dev)
    flat.get_store([p for p in task.parameters() if p.requires_grad])
    return task, sd


def main():
    which = [int(a) for a in sys.argv[1:]] or list(range(len(CASES)))
    for ci in which:
        rv, (cs, lcf) = CASES[ci]
        cfg = bench.c3_config(500)
        cfg["encoder"]["config"]["chunk_size"] = [cs]
        cfg["encoder"]["config"]["left_context_frames"] = [lcf]
        lcc = -1 if cs < 0 else max(1, lcf // cs)
        batch = bench.make_batch(0, 2, 10.0, 50, 500, dev)
        batch["pcm_length"][1] = 131000
        batch["label_length"][1] = 37
        ref_grads = None
        real_random = random.random
        for pol in POLICIES:
            set_policy(pol)
            random.random = real_random
            task, sd = build(cfg)
            task.eval()
            with torch.no_grad():
                feat, feat_len = task.features(batch)
            task.train()
            for mod in task.modules():
                if mod.__class__.__name__ == "CompactRelPositionalEncoding":
                    mod.dropout.p = 0.0
            fb = {"feat": feat, "feat_length": feat_len, "label": batch["label"], "label_length": batch["label_length"]}
            random.random = lambda: rv
            # warm the plan tables of this policy on a second task object (see the test)
            st = random.getstate()
            warm, _ = build(cfg)
            warm.train()
            warm.training_step(fb, 0).backward()
            torch.cuda.synchronize()
            del warm
            random.setstate(st)
            n0 = list(zip_native.CALLS)
            torch.manual_seed(7)
            loss = task.training_step(fb, 0)
            loss.backward()
            torch.cuda.synchronize()
            native = zip_native.CALLS[0] - n0[0]
            if ref_grads is None:
                enc_sd = {k[len("_encoder.encoder."):]: v for k, v in sd.items() if k.startswith("_encoder.encoder.")}
                lab, lab_len = batch["label"].cpu(), batch["label_length"].cpu()
                torch.manual_seed(7)
                yo, ylo = Z.zipformer_forward(enc_sd, bench._zcfg(cfg["encoder"]["config"]), feat.cpu(),
                                              feat_len.cpu(), Z.Ctl(True, lambda: rv, pos_dropout=0.0), cs, lcc)
                po = H.stateless_predictor(sd, "_predictor.predictor.", lab, 5)
                am, lm = H.joiner_projections(sd, "_joiner.", yo, po)
                lo, bo, ro, so = K2.joiner_pruned(am, lm, lab, lab_len, ylo, 5)
                pro = K2.rnnt_loss_pruned(lo, lab, ro, 0, bo)
                ref = H.pruned_rnnt_task_loss(so, pro, 0.5, 0.5)
                ref.backward()
                ref_loss = float(ref)
                ref_grads = {n: (torch.zeros_like(sd[n]) if sd[n].grad is None else sd[n].grad.clone())
                             for n, _ in task.named_parameters()}
            errs = []
            for n, p in task.named_parameters():
                r = ref_grads[n]
                g = torch.zeros_like(r) if p.grad is None else p.grad.detach().cpu()
                errs.append(((g - r).abs().max().item() / (r.abs().max().item() + 1e-6), n))
            errs.sort(reverse=True)
            over = sum(e > 5e-3 for e, _ in errs)
            print(f"case {ci} rv {rv} chunk {cs}/{lcf} policy FDWS={pol} native {native}/12: loss rel err "
                  f"{abs(float(loss) - ref_loss) / abs(ref_loss):.2e}; params over 5e-3: {over}; worst: "
                  + "; ".join(f"{e:.2e} {n.replace('_encoder.encoder.', '')}" for e, n in errs[:4]), flush=True)
            random.random = real_random
            del task


if __name__ == "__main__":
    main()
