"""Where a step ends: from a rocprofv3 kernel trace of bench.py, for each of the last steps (delimited
by the optimizer's scaled_adam_apply kernel) the time between the LAST kernel of the main queue's
backward pass and the first optimizer kernel -- i.e. how long the optimizer waits for the side
stream's weight gradients -- and which side-stream kernels run in that window.
usage: python tools/prof_tail.py TRACE.csv [STEPS]"""
import collections
import csv
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:n.index("(")] if "(" in n and not n.startswith("Cijk") else n[:60]


def main():
    rows = [(r["Kernel_Name"], r.get("Queue_Id", "0"), int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
            for r in csv.DictReader(open(sys.argv[1]))]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rows.sort(key=lambda r: r[2])
    byq = collections.Counter(r[1] for r in rows)
    main_q = byq.most_common(1)[0][0]
    marks = [i for i, r in enumerate(rows) if "scaled_adam_apply" in r[0]]
    for k in range(len(marks) - steps, len(marks)):
        lo = rows[marks[k - 1]][3] if k > 0 else rows[0][2]
        hi = rows[marks[k]][2]
        seg = [r for r in rows if lo <= r[2] < hi]
        # optimizer prologue on the main queue: the clip / seg_stats / coef kernels just before apply
        main = [r for r in seg if r[1] == main_q]
        side = [r for r in seg if r[1] != main_q]
        opt_names = ("seg_stats", "scaled_adam", "param_grad_commit", "clip")
        bw = [r for r in main if not any(o in r[0] for o in opt_names)]
        last_bw = max(r[3] for r in bw)
        first_opt = min((r[2] for r in main if any(o in r[0] for o in opt_names) and r[2] >= last_bw), default=hi)
        last_side = max((r[3] for r in side), default=lo)
        tail = [r for r in side if r[3] > last_bw]
        print(f"step {k}: length {(hi - lo) / 1e6:.2f} ms; main backward ends at {(last_bw - lo) / 1e6:.2f} "
              f"({short(bw[-1][0])}), side stream ends at {(last_side - lo) / 1e6:.2f}, first optimizer kernel "
              f"at {(first_opt - lo) / 1e6:.2f}: the optimizer waits {(first_opt - last_bw) / 1e6:.3f} ms")
        agg = collections.defaultdict(float)
        for r in tail:
            agg[short(r[0])] += (r[3] - max(r[2], last_bw)) / 1e6
        for n, ms in sorted(agg.items(), key=lambda kv: -kv[1])[:6]:
            print(f"      after the main stream's last backward kernel: {ms:.3f} ms of {n}")
        # the last 8 main-queue kernels of backward with their gaps
        for a, b in zip(bw[-9:-1], bw[-8:]):
            print(f"      main tail: {short(b[0])[:48]:48s} start +{(b[2] - lo) / 1e6:.2f} ms, gap before {(b[2] - a[3]) / 1e3:.0f} us, runs {(b[3] - b[2]) / 1e3:.0f} us")


if __name__ == "__main__":
    main()
