"""GPU: bench.py's JSON line honours the driver's contract and its roofline table is credible."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg,arith", [("C3", None), ("C2", None), ("C3", "bf16x3/6"), ("C3", "bf16x2/3")])
def test_bench_line_contract_and_roofline_rows(dev, cfg, arith):
    """arith: S2T_GEMM_ARITH of the run (None = the built-in default): the line states the arithmetic
    (config.gemm_arith) and the roofline block's bf16 ceiling follows it (2.5 PF / 6 or / 3)."""
    env = dict(os.environ)
    env.pop("S2T_GEMM_ARITH", None)
    if arith:
        env["S2T_GEMM_ARITH"] = arith
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "3",
                        "--warmup", "2", "--no-cpu-baseline", "--profile-steps", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "audio-seconds/sec" and d["n_gpus"] == 1 and d["steps"] == 3
    assert d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    pol = d["config"]["gemm_arith"]            # one name, or "F .., D .., W .." when those classes differ
    assert pol in ("bf16x3/6", "bf16x2/3") or pol.startswith("F bf16x")
    cls = d["config"]["gemm_arith_classes"]
    assert set(cls) == set("FDWS") and all(v in ("bf16x3/6", "bf16x2/3") for v in cls.values())
    if arith:                                  # S2T_GEMM_ARITH sets every class, the statistics included
        assert pol == arith and set(cls.values()) == {arith}
    else:                                      # the built-in policy: the statistics on six products
        assert pol == "bf16x2/3" and cls["S"] == "bf16x3/6"
    products = 3.0 if ("F bf16x2" in pol or "D bf16x2" in pol or pol == "bf16x2/3") else 6.0
    assert isinstance(d["cpu_baseline"], dict) and d["cpu_baseline"]["kind"] == "port"
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["achieved"] and 0 < rf["frac"] <= 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["launches"] >= 1 and rf["avg_launch_ms"] > 0
    # every row of the per-entry table must stay below its roofline: a fraction above 1 means the
    # algorithmic bytes / flops were noted under the wrong entry (round-2 verdict, item 9)
    assert len(rf["kernels"]) >= 10
    # (an entry that multiplies on the bf16 matrix cores with the six-product split may exceed the
    # f32-MFMA peak the table quotes -- the implicit-operand conv GEMM does, 158 TFLOP/s -- and is
    # held to the six-product ceiling instead)
    for row in rf["kernels"]:
        for key in ("frac_hbm", "frac_mfma"):
            if row.get(key) is not None:
                top = 1.0 if (key == "frac_hbm" or "frac_bf16_ceiling" not in row) else 2500.0 / products / 157.3
                assert 0 <= row[key] <= top, (row["entry"], key, row[key])
        if "frac_bf16_ceiling" in row:
            assert 0 <= row["frac_bf16_ceiling"] <= 1.0, (row["entry"], row["frac_bf16_ceiling"])
    # the step's own fraction of the f32 MFMA peak
    assert 0 < rf["step_mfma"]["frac"] < 1.0
    # no blind rows -- every entry that costs >= 0.1 ms per step carries its
    # algorithmic bytes (and flops where it multiplies), so the table prices all of the step
    blind = [(r["entry"], round(r["ms_per_step"], 3)) for r in rf["kernels"]
             if r["ms_per_step"] >= 0.1 and "frac_hbm" not in r and "frac_mfma" not in r]
    assert not blind, blind
    # entries on the bf16 matrix cores also report against the ceiling of their arithmetic
    # (2.5 PF / 6 products, or / 3), and the line's own block says which
    for row in rf["kernels"]:
        if row["entry"] in ("s2t_gemm_x3p", "s2t_gemm_tn_grouped") and "frac_mfma" in row:
            assert 0 < row["frac_bf16_ceiling"] < row["frac_mfma"]
    if rf["kernel"] == "s2t_gemm_x3p":
        assert rf["gemm_arith"] == d["config"]["gemm_arith"]
        assert abs(rf["bf16_ceiling_tflops"] - 2500.0 / products) < 1e-6
        assert 0 < rf["algorithmic_bytes_min_per_launch"] <= rf["algorithmic_bytes_per_launch"]
    assert d["config"]["gemm_paths"]["aten_fallbacks"] == 0
