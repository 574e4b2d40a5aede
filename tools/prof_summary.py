"""Summarise a rocprofv3 --kernel-trace --stats (csv) output directory into profiles/."""
import collections, csv, glob, os, sys


def main(src, tag, steps, ms_per_step):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out, exist_ok=True)
    stats = glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows[:80]:
            w.writerow(r)
    trace = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)[0]
    tr = list(csv.DictReader(open(trace)))
    tend = max(int(r["End_Timestamp"]) for r in tr)
    win = steps * ms_per_step * 1e6
    sel = [r for r in tr if int(r["Start_Timestamp"]) >= tend - win]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        a = agg[r["Kernel_Name"]]
        a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[1] += 1
    tot = sum(a[0] for a in agg.values())

    def klass(k):
        if k.startswith("Cijk_") or "rocblas" in k.lower() or "hipblaslt" in k.lower():
            return "GEMM library (hipBLASLt/rocBLAS)"
        if "at::native" in k or "at::cuda" in k or k.startswith("void at::"):
            return "ATen elementwise/reduce"
        if "rocclr" in k or "hipMemcpy" in k or "copyBuffer" in k or "fillBuffer" in k:
            return "copy/fill"
        if ("miopen" in k.lower() or "Im2d2Col" in k or "Col2Im" in k or k.startswith("igemm_")
                or k.startswith("batched_transpose") or k.startswith("SubTensorOp")):
            return "MIOpen"
        if k in ("attn_fwd", "bwd_kernel_dk_dv", "bwd_kernel_dq", "bwd_preprocess"):
            return "AOTriton (torch SDPA)"
        if "ccl" in k.lower():
            return "RCCL"
        return "hand-written HIP (libs2t_mi355)"

    # timeline of the window: union of all kernels' intervals (= time with at least one kernel on
    # the chip), and the gaps between consecutive kernels of the busiest queue (the main stream)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
    union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    union += cur_e - cur_s
    span = iv[-1][1] - iv[0][0]
    qkey = "Queue_Id" if "Queue_Id" in sel[0] else ("Stream_Id" if "Stream_Id" in sel[0] else None)
    tl = [f"# timeline: window {span/steps/1e6:.2f} ms/step, some kernel running {union/steps/1e6:.2f} ms/step, "
          f"chip idle {(span-union)/steps/1e6:.2f} ms/step"]
    if qkey:
        byq = collections.defaultdict(list)
        for r in sel:
            byq[r[qkey]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1]))[:3]:
            lst.sort()
            busy = sum(b - a for a, b in lst)
            gaps = sorted(max(0, lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1))
            short = [g for g in gaps if g < 100000]
            tl.append(f"# {qkey} {q}: {len(lst)/steps:.0f} launches/step, busy {busy/steps/1e6:.2f} ms/step, "
                      f"gaps < 100 us: median {short[len(short)//2]/1e3:.1f} us, sum {sum(short)/steps/1e6:.2f} ms/step")
    cls = collections.defaultdict(lambda: [0, 0])
    for k, (d, n) in agg.items():
        c = cls[klass(k)]
        c[0] += d
        c[1] += n
    with open(os.path.join(out, f"{tag}_timed_region.txt"), "w") as f:
        f.write(f"# kernels whose start lies in the last {steps} steps ({ms_per_step:.1f} ms each) of the trace\n")
        f.write(f"# GPU busy {tot/steps/1e6:.2f} ms/step, {len(sel)/steps:.0f} launches/step\n")
        for line in tl:
            f.write(line + "\n")
        for c, (d, n) in sorted(cls.items(), key=lambda x: -x[1][0]):
            f.write(f"# class {c:36s} {d/steps/1e6:8.3f} ms/step {100*d/tot:5.1f}% {n/steps:8.1f} launches/step\n")
        for k, (d, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:70]:
            f.write(f"{d/steps/1e6:8.3f} ms/step {100*d/tot:5.1f}% {n/steps:8.1f} calls/step "
                    f"{d/n/1e3:9.1f} us avg  {k[:140]}\n")
        if qkey:
            # what the second-busiest queue (the side stream) runs
            qs = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
            for r in sel:
                a = qs[r[qkey]][r["Kernel_Name"]]
                a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                a[1] += 1
            order = sorted(qs.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))
            for q, ks in order[1:2]:
                f.write(f"# ---- {qkey} {q} (side stream) by kernel\n")
                for k, (d, n) in sorted(ks.items(), key=lambda x: -x[1][0])[:25]:
                    f.write(f"#   {d/steps/1e6:8.3f} ms/step {n/steps:8.1f} calls/step {d/n/1e3:9.1f} us avg  {k[:120]}\n")
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4]))
