"""One step of a rocprofv3 kernel trace of bench.py as a launch-by-launch listing: queue, start offset
(us from the step's first kernel), duration, gap to the previous kernel of the same queue, short name
and grid.  Steps are delimited by the optimizer's scaled_adam_apply kernel; the last complete one is
written.  usage: python tools/prof_sequence.py TRACE.csv OUT.txt"""
import collections
import csv
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("Cijk"):
        return n[:48]
    if "at::native" in n:
        i = n.find("at::native::", n.find("<") + 1)
        tail = n[i + 12:i + 60] if i > 0 else ""
        return "aten:" + n[12:40] + "|" + tail
    return n[:n.index("(")] if "(" in n else n[:70]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"],
                     r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "scaled_adam_apply" in r[3]]
    lo, hi = marks[-2] + 1, marks[-1] + 1
    seg = rows[lo:hi]
    t0 = seg[0][0]
    mainq = collections.Counter(r[2] for r in seg).most_common(1)[0][0]
    last_end = {}
    with open(sys.argv[2], "w") as f:
        f.write(f"# {len(seg)} launches, {(seg[-1][1] - t0) / 1e6:.2f} ms; main queue = {mainq}; columns: queue start_us dur_us gap_us name grid/wg\n")
        for s, e, q, n, g, w in seg:
            gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
            last_end[q] = e
            f.write(f"{'M' if q == mainq else 'S'} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:7.1f}  {short(n)}  {g}/{w}\n")


if __name__ == "__main__":
    main()
