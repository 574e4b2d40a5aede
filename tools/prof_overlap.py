"""Per-queue view of a rocprofv3 kernel trace: for the kernels of the busiest queue (the main
stream), how much of their run time has a kernel of another queue on the chip at the same time.
usage: python tools/prof_overlap.py TRACE.csv STEPS MS_PER_STEP [OUT.csv]
With OUT.csv the selected window is written back in compact form (name, queue, start, end, grid)."""
import collections, csv, sys, bisect


def load(path, steps, ms):
    tr = list(csv.DictReader(open(path)))
    if "Start_Timestamp" in tr[0]:
        def grid(r):
            if "Grid_Size_X" in r:            # work-items per dimension / workgroup size
                g = [int(r.get("Grid_Size_" + a, 1) or 1) for a in "XYZ"]
                w = [int(r.get("Workgroup_Size_" + a, 1) or 1) for a in "XYZ"]
                return "%dx%d" % (g[0] * g[1] * g[2] // max(1, w[0] * w[1] * w[2]), w[0] * w[1] * w[2])
            return r.get("Grid_Size", "")
        rows = [(r["Kernel_Name"], r.get("Queue_Id", "0"), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                 grid(r)) for r in tr]
    else:
        rows = [(r["name"], r["queue"], int(r["start"]), int(r["end"]), r["grid"]) for r in tr]
    tend = max(r[3] for r in rows)
    win = steps * ms * 1e6
    return [r for r in rows if r[2] >= tend - win]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:n.index("(")] if "(" in n and not n.startswith("Cijk") else n[:60]


def main():
    path, steps, ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    rows = load(path, steps, ms)
    if len(sys.argv) > 4:
        with open(sys.argv[4], "w") as f:
            w = csv.writer(f)
            w.writerow(["name", "queue", "start", "end", "grid"])
            for r in rows:
                w.writerow([short(r[0]), r[1], r[2], r[3], r[4]])
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[1]].append(r)
    main_q = max(byq, key=lambda q: len(byq[q]))
    others = sorted((r[2], r[3]) for q, rs in byq.items() if q != main_q for r in rs)
    # merge the other queues' intervals
    merged = []
    for a, b in others:
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    starts = [m[0] for m in merged]

    def overlap(a, b):
        i = max(0, bisect.bisect_right(starts, a) - 1)
        tot = 0
        while i < len(merged) and merged[i][0] < b:
            tot += max(0, min(b, merged[i][1]) - max(a, merged[i][0]))
            i += 1
        return tot

    agg = collections.defaultdict(lambda: [0, 0, 0, 0, 0, 0])   # n, time, overlapped time, n_alone, t_alone, n_ov/t_ov
    for r in byq[main_q]:
        d = r[3] - r[2]
        ov = overlap(r[2], r[3])
        a = agg[short(r[0])]
        a[0] += 1
        a[1] += d
        a[2] += ov
        if ov < 0.1 * d:
            a[3] += 1
            a[4] += d
        elif ov > 0.9 * d:
            a[5] += d
    tot = sum(a[1] for a in agg.values())
    tov = sum(a[2] for a in agg.values())
    side = sum(b - a for a, b in merged)
    print(f"main queue {main_q}: {len(byq[main_q]) / steps:.0f} launches/step, busy {tot / steps / 1e6:.2f} ms/step, "
          f"of which {tov / steps / 1e6:.2f} ms with another queue's kernel on the chip; other queues busy "
          f"{side / steps / 1e6:.2f} ms/step (union)")
    print(f"{'ms/step':>8} {'calls':>6} {'avg us':>7} {'ovl %':>6} {'alone: n':>8} {'avg us':>7}   kernel")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{a[1] / steps / 1e6:8.3f} {a[0] / steps:6.1f} {a[1] / a[0] / 1e3:7.1f} {100.0 * a[2] / a[1]:6.1f} "
              f"{a[3] / steps:8.1f} {(a[4] / a[3] / 1e3) if a[3] else 0:7.1f}   {k}")


if __name__ == "__main__":
    main()
