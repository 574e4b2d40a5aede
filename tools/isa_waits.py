"""Static scan of the gfx950 ISA of every kernel (hipcc -S output under /tmp/isa, see the usage
line) for the patterns that defeat load look-ahead: s_waitcnt vmcnt(0) next to many loads,
flat_load, scratch traffic -- weighted by the kernel's time in a trace written by
tools/prof_overlap.py.
usage: (cd speech2text_amd/csrc && for f in *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 \
        -S --cuda-device-only -o /tmp/isa/$f.s $f; done); python tools/isa_waits.py [TRACE.csv]"""
import collections, csv, glob, re, subprocess, sys

tshare = collections.defaultdict(float)
if len(sys.argv) > 1:
    for r in csv.DictReader(open(sys.argv[1])):
        tshare[r["name"]] += (int(r["end"]) - int(r["start"])) / 5 / 1e6
names, recs = [], []
for f in glob.glob("/tmp/isa/*.s"):
    s = open(f).read()
    for m in re.finditer(r"^(_Z\S+):\s*;\s*@\S+\n", s, re.M):
        i = m.end()
        j = s.find(".Lfunc_end", i)
        body = s[i:j]
        if j < 0 or "s_endpgm" not in body:
            continue
        w = re.findall(r"vmcnt\((\d+)\)", body)
        recs.append((m.group(1), len(re.findall(r"\b(?:global_load|flat_load|buffer_load)", body)),
                     len(re.findall(r"\bflat_load", body)), len(re.findall(r"\bscratch_", body)),
                     sum(1 for x in w if x == "0"), len(w)))
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in recs), capture_output=True, text=True).stdout.split("\n")
out = []
for (mn, loads, flat, scr, w0, wn), d in zip(recs, dem):
    short = re.sub(r"^void ", "", d.replace("(anonymous namespace)::", ""))
    short = re.sub(r"\(.*", "", short) if not short.startswith("(") else short
    out.append((tshare.get(short, 0.0), short, loads, flat, scr, w0, wn))
out.sort(reverse=True)
print("%7s %-62s %5s %4s %4s  %s" % ("ms/step", "kernel (instantiations that ran in the trace)", "loads", "flat", "scr", "vmcnt(0)/all"))
for t, short, loads, flat, scr, w0, wn in out:
    if t > 0:
        print("%7.3f %-62s %5d %4d %4d  %3d/%3d" % (t, short[:62], loads, flat, scr, w0, wn))
