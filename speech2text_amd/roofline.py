"""Roofline bookkeeping for bench.py: algorithmic bytes of the dominant hand-written kernel
(formulas in DESIGN.md / SURVEY.md section 8d) divided by its live-measured launch time, against the
MI355X HBM peak (8 TB/s spec, /opt/skills/guides/MI355X_MICROARCH.md)."""
import json
import os

HBM_PEAK_GBS = 8000.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(entry, prof, args):
    if not prof or not prof.get("launches"):
        return {"bound": "hbm", "kernel": entry, "achieved": None, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": None, "traffic": None,
                "note": "kernel was not launched in the timed region"}
    per_launch = prof["algo_bytes"] / prof["launches"]
    achieved = prof["algo_bytes"] / (prof["total_ms"] * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(entry, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": entry, "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "launches": prof["launches"], "avg_launch_ms": prof["avg_ms"],
            "algorithmic_bytes_per_launch": per_launch}
