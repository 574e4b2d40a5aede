"""CPU: bench.py's CLI contract -- the oracle-only cpu_baseline leg runs, the roofline kernel
names are real C entry points, and `--gpus 2` starts its own two ranks (gloo rehearsal)."""
import json
import os
import subprocess
import sys

import torch

import bench
from speech2text_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_c3():
    cfg = bench.c3_config(64)
    cfg["encoder"]["config"].update({"downsampling_factor": [1, 2], "num_encoder_layers": [1, 1],
                                     "feedforward_dim": [96, 128], "encoder_dim": [48, 64],
                                     "encoder_unmasked_dim": [32, 48], "num_heads": [4, 4],
                                     "query_head_dim": 8, "value_head_dim": 4, "pos_dim": 16,
                                     "cnn_module_kernel": [15, 7]})
    cfg["predictor"]["config"].update({"output_dim": 64, "symbol_embedding_dim": 32})
    cfg["joiner"].update({"input_dim": 64})
    return cfg


def test_cpu_baseline_runs_on_oracle_only(monkeypatch):
    from speech2text_amd import zip_kernels as zk
    from speech2text_amd.build_task import TaskFactory

    def boom(*a, **k):
        raise AssertionError("cpu_baseline must not call product kernels")

    cfg = _tiny_c3()
    torch.manual_seed(0)
    task = TaskFactory.get("Pruned_Rnnt")(cfg)
    sd = task.state_dict()
    for name in ("linear", "swoosh_forward", "bias_norm", "relpos_attention_weights"):
        monkeypatch.setattr(zk, name, boom)
    r = bench.cpu_baseline(cfg, sd, seconds=1.5, batch=2, n_labels=4, vocab=64, steps=3, warmup=1)
    assert r["kind"] == "port" and r["cores"] >= 1 and r["value"] > 0
    assert "3 timed steps" in r["sample"]


def test_cpu_baseline_c2_runs():
    from speech2text_amd.build_task import TaskFactory
    cfg = bench.c2_config(32, layers=2)
    cfg["encoder"]["config"].update({"input_dim": 32, "ffn_dim": 64, "output_dim": 32})
    cfg["decoder"]["config"].update({"input_dim": 32})
    torch.manual_seed(0)
    task = TaskFactory.get("CTC")(cfg)
    r = bench.cpu_baseline_c2(cfg, task.state_dict(), seconds=1.5, batch=2, n_labels=4, vocab=32,
                              steps=1, warmup=0)
    assert r["value"] > 0


def test_roofline_kernel_names_are_entry_points():
    protos = _native.parse_header()
    args = bench.parse_args([])
    assert args.gpus == 1 and args.steps >= 1 and args.config == "C3"
    assert args.roofline_kernel == "auto" or args.roofline_kernel in protos
    assert "s2t_relpos_attn_fwd" in protos          # the fallback used when "auto" finds nothing
    try:
        _native.profile_begin("relpos_attn_weights_fwd")
    except ValueError:
        pass
    else:
        raise AssertionError("profile_begin accepted a name that is not an entry point")


def test_gpus2_self_launch_gloo_rehearsal():
    env = dict(os.environ, MASTER_PORT="29731")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
                        "3", "--warmup", "1", "--launcher-selftest"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["backend"] == "gloo"
