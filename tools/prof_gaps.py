"""Main-stream gaps of a compact kernel trace (tools/prof_overlap.py OUT.csv): every stretch of
>= MIN us with no main-queue kernel running, and what the other queues run meanwhile.
usage: python tools/prof_gaps.py TRACE.csv [MIN_US]"""
import collections, csv, sys


def main():
    rows = [(r["name"], r["queue"], int(r["start"]), int(r["end"])) for r in csv.DictReader(open(sys.argv[1]))]
    mn = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
    cnt = collections.Counter(r[1] for r in rows)
    mainq = cnt.most_common(1)[0][0]
    main = sorted((r for r in rows if r[1] == mainq), key=lambda r: r[2])
    side = sorted((r for r in rows if r[1] != mainq), key=lambda r: r[2])
    t0 = main[0][2]
    tot = 0.0
    end = main[0][3]
    prev = main[0]
    for r in main[1:]:
        if r[2] - end >= mn * 1e3:
            gap = (r[2] - end) / 1e3
            tot += gap
            names = collections.Counter()
            for s in side:
                ov = min(s[3], r[2]) - max(s[2], end)
                if ov > 0:
                    names[s[0]] += ov / 1e3
            what = ", ".join(f"{k} {v:.0f}" for k, v in names.most_common(4)) or "(chip idle)"
            print(f"t={(end - t0) / 1e6:8.2f} ms  gap {gap:7.1f} us  after {prev[0][:40]:40s} before {r[0][:40]:40s} | side: {what}")
        if r[3] > end:
            end = r[3]
            prev = r
    print(f"total main-queue gaps >= {mn} us: {tot / 1e3:.2f} ms over the window")


main()
