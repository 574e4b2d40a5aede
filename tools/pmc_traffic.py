"""HBM traffic per launch of every hand-written entry point, from two rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as MI355X_MICROARCH.md prescribes) over
the same bench command.  Units / gfx950 corrections per that guide's HBM section: both counters
are in KiB; FETCH_SIZE counts a wide coalesced streaming read at exactly half its bytes, so the
read side is reported raw and doubled (`hbm_bytes_per_launch` uses the doubled value);
WRITE_SIZE is exact for 16-byte stores and float atomics.

    python tools/pmc_traffic.py FETCH_DIR WRITE_DIR [out.json]
"""
import collections
import csv
import glob
import json
import os
import sys

# kernel-name substring -> (C entry point, is_primary): the primary kernel counts the launches
KERNELS = [
    ("x3p_db_kernel", "s2t_gemm_x3p", True), ("x3p_kernel", "s2t_gemm_x3p", True),
    ("x3p_split_kernel", "s2t_x3p_split", True),
    ("dwconv2d_roll_kernel<7, 7, 2>", "s2t_dwconv2d_nhwc_wgrad", True),
    ("dwconv2d_wgrad_ring_kernel", "s2t_dwconv2d_nhwc_wgrad", True),
    ("gemm_tn_grouped_w_kernel", "s2t_gemm_tn_grouped", True), ("gemm_tn_w_kernel", "s2t_gemm_f32", True),
    ("x3p_dma_kernel", "s2t_gemm_x3p", True),
    ("dwconv2d_roll_kernel", "s2t_dwconv2d_nhwc_fwd", True),
    ("attn_fwd_mfma_kernel", "s2t_relpos_attn_fwd", True), ("attn_fwd_kernel", "s2t_relpos_attn_fwd", True),
    ("attn_bwd_q", "s2t_relpos_attn_bwd", True), ("attn_bwd_k", "s2t_relpos_attn_bwd", False),
    ("dpos_reduce", "s2t_relpos_attn_bwd", False),
    ("attn_apply_kernel", "s2t_attn_apply", True), ("attn_apply4_kernel", "s2t_attn_apply", True),
    ("downsample_bwd", "s2t_downsample_bwd", True), ("bypass_up_bwd", "s2t_bypass_up_bwd", True),
    ("dropout_add_kernel", "s2t_dropout_add", True), ("silu_drop_fwd_kernel", "s2t_silu_fwd", True),
    ("silu_drop_bwd_kernel", "s2t_silu_bwd", True),
    ("gemm_kernel", "s2t_gemm_f32", True), ("gemm_tn_grouped_kernel", "s2t_gemm_tn_grouped", True),
    ("wgrad_kernel", "s2t_linear_wgrad", True), ("wgrad_reduce_kernel", "s2t_linear_wgrad", False),
    ("zipconv_fwd_kernel", "s2t_zipconv_fwd", True),
    ("zipconv_bwd_data_kernel", "s2t_zipconv_bwd", True), ("zipconv_bwd_w_kernel", "s2t_zipconv_bwd", False),
    ("zipconv_reduce_w_kernel", "s2t_zipconv_bwd", False),
    ("swoosh_fwd_kernel", "s2t_swoosh_fwd", True), ("swoosh_bwd_kernel", "s2t_swoosh_bwd", True),
    ("biasnorm_fwd_kernel", "s2t_biasnorm_fwd", True), ("biasnorm_bwd_kernel", "s2t_biasnorm_bwd", True),
    ("col_stats_kernel", "s2t_balancer_bwd", True),
    ("balancer_apply_fused_kernel", "s2t_balancer_bwd", False),
    ("bypass_fwd", "s2t_bypass_fwd", True), ("bypass_bwd", "s2t_bypass_bwd", True),
    ("nonlin_gate_fwd", "s2t_nonlin_gate_fwd", True), ("nonlin_out_bwd", "s2t_nonlin_out_bwd", True),
    ("whiten_metric_kernel", "s2t_whiten_metric", True), ("col2im3x3_kernel", "s2t_col2im3x3_nhwc", True),
    ("attn_delta_pairs_kernel", "s2t_attn_delta_pairs", True),
    ("whiten_apply_kernel", "s2t_whiten_apply", True), ("sumsq2_kernel", "s2t_whiten_apply", False),
    ("mi_fwd_kernel", "s2t_mutual_info_fwd", True), ("mi_bwd_kernel", "s2t_mutual_info_bwd", True),
    ("pruned_fwd", "s2t_rnnt_pruned_fwd", True), ("pruned_bwd", "s2t_rnnt_pruned_bwd", True),
    ("simple_pxpy", "s2t_rnnt_simple_pxpy", True), ("simple_bwd", "s2t_rnnt_simple_bwd", True),
    ("row_exp", "s2t_rnnt_row_exp", True),
    ("ctc_alpha_beta_kernel", "s2t_ctc_loss_fwd_bwd", True), ("ctc_lse_gather_kernel", "s2t_ctc_loss_fwd_bwd", False),
    ("ctc_grad_kernel", "s2t_ctc_loss_fwd_bwd", False),
    ("fbank", "s2t_fbank_f32", True),
    ("seg_stats_kernel", "s2t_seg_stats", True), ("scaled_adam_apply_kernel", "s2t_scaled_adam_apply", True),
    ("dwconv2d_kernel", "s2t_dwconv2d_nhwc_fwd", True), ("dwconv2d_wgrad_kernel", "s2t_dwconv2d_nhwc_wgrad", True),
    ("dwconv2d_wreduce_kernel", "s2t_dwconv2d_nhwc_wgrad", False),
    ("smoothed_nll_fwd_kernel", "s2t_smoothed_nll_fwd", True), ("smoothed_nll_bwd_kernel", "s2t_smoothed_nll_bwd", True),
    ("bestrq", "s2t_bestrq_labels", True),
    ("Cijk_", "s2t_linear_lt", True),                       # hipBLASLt kernels (forward / dgrad GEMMs)
    ("layernorm_fwd_kernel", "s2t_layernorm_fwd", True), ("layernorm_bwd_kernel", "s2t_layernorm_bwd", True),
    ("silu_fwd_kernel", "s2t_silu_fwd", True), ("silu_bwd_kernel", "s2t_silu_bwd", True),
    ("mhsa_fwd_kernel", "s2t_mhsa_fwd", True), ("mhsa_bwd_q_kernel", "s2t_mhsa_bwd", True),
    ("mhsa_bwd_kv_kernel", "s2t_mhsa_bwd", False),
    ("bn_silu_apply_kernel", "s2t_bn_silu_fwd", True), ("bn_stats_kernel<0>", "s2t_bn_silu_fwd", False),
    ("bn_silu_bwd_kernel", "s2t_bn_silu_bwd", True), ("bn_stats_kernel<1>", "s2t_bn_silu_bwd", False),
    ("conv1_relu_fwd_kernel", "s2t_conv1_relu_fwd", True), ("conv1_relu_wgrad_kernel", "s2t_conv1_relu_wgrad", True),
    ("lnlstm_fwd_kernel", "s2t_lnlstm_fwd", True), ("lnlstm_bwd_kernel", "s2t_lnlstm_bwd", True),
    ("adam_apply_kernel", "s2t_adam_apply", True),
]


import re
_GEMM = re.compile(r"gemm_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false)>")


def entry_of(kname):
    # our NT / NN / TN kernel serves several entry points: the plan cache's launches
    # (s2t_linear_lt, modes 0 / 1), the implicit-im2col convs (PATCH) and the single TN products
    m = _GEMM.search(kname)
    if m:
        mode, patch = int(m.group(3)), m.group(6) == "true"
        if patch:
            return ("gemm_kernel", "s2t_conv3x3_gemm", True)
        if mode in (0, 1):
            return ("gemm_kernel", "s2t_linear_lt", True)
        return ("gemm_kernel", "s2t_gemm_f32", True)
    best = None
    for sub, entry, prim in KERNELS:
        if sub in kname and (best is None or len(sub) > len(best[0])):
            best = (sub, entry, prim)
    return best


def collect(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    rows = []
    for f in files:
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    # only the LAST step of the run: between the last two optimizer-apply dispatches.  The first
    # step also holds the plan cache's tuning launches (thousands of GEMMs with a different shape
    # mix), which made the per-launch mean of s2t_linear_lt incomparable with the in-step figure.
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        marks = [i for i, r in enumerate(rows) if "adam_apply_kernel" in r["Kernel_Name"]]
        if len(marks) >= 2:
            rows = rows[marks[-2] + 1:marks[-1] + 1]
    for f in [None]:
        for r in rows:
            e = entry_of(r["Kernel_Name"])
            if e is None:
                continue
            tot[e[1]] += float(r["Counter_Value"])
            if e[2]:
                cnt[e[1]] += 1
    return tot, cnt


def main(fetch_dir, write_dir, out, bench_json=None):
    ft, fc = collect(fetch_dir, "FETCH_SIZE")
    wt, wc = collect(write_dir, "WRITE_SIZE")
    res = {}
    for e in sorted(set(ft) | set(wt)):
        n = fc.get(e) or wc.get(e)
        if not n:
            continue
        fetch_raw = 1024.0 * ft.get(e, 0.0) / n
        write = 1024.0 * wt.get(e, 0.0) / max(1, wc.get(e, n))
        res[e] = {"launches_sampled": n, "fetch_bytes_raw_per_launch": fetch_raw,
                  "fetch_bytes_x2_per_launch": 2.0 * fetch_raw, "write_bytes_per_launch": write,
                  "hbm_bytes_per_launch": 2.0 * fetch_raw + write,
                  "note": "mean over the launches of the last bench step; FETCH_SIZE raw and x2 "
                          "(gfx950 counts 128-byte requests at 64 bytes: calibrated per access "
                          "width with tools/probes/fetch_probe.hip, profiles/r03_fetch_calibration.txt"
                          "), WRITE_SIZE exact"}
    if bench_json and os.path.exists(bench_json):
        # the bench line of the PMC pass itself (--steps 1 --sample-every 1): algorithmic bytes of
        # exactly the launches counted above
        try:
            r = json.loads(open(bench_json).read().strip().splitlines()[-1])["roofline"]
            if r.get("kernel") in res and r.get("algorithmic_bytes_per_launch"):
                res[r["kernel"]]["algorithmic_bytes_per_launch"] = r["algorithmic_bytes_per_launch"]
                res[r["kernel"]]["algorithmic_launches"] = r.get("launches")
        except Exception as e:                       # noqa: BLE001
            print("bench line not usable:", e)
    json.dump(res, open(out, "w"), indent=1)
    for e, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print(f"{e:28s} n={v['launches_sampled']:5d} fetch(x2) {v['fetch_bytes_x2_per_launch']/1e6:9.2f} MB "
              f"write {v['write_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    main(sys.argv[1], sys.argv[2],
         sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, "profiles", "roofline_traffic.json"),
         sys.argv[4] if len(sys.argv) > 4 else None)
