"""(CPU) Fold a rocprofv3 kernel trace of `tools/train_probe.py c2 f32 N` into a per-stream picture of the LAST training step:
which queue each kernel ran on, busy time per queue, idle gaps of the main queue, and the kernels that ran while another
queue was also busy (slowed by sharing the chip).  The step boundary is the mse_kernel launch (one per step).
  python tools/train_timeline.py <kernel_trace.csv> [out.txt]"""
import collections
import csv
import sys


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:58]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
    ks.sort()
    marks = [i for i, k in enumerate(ks) if k[2].startswith("mse_kernel")]
    if len(marks) < 3:
        print("fewer than three steps in the trace", file=out)
        return
    a, b = marks[-2], marks[-1]            # one whole step: from one loss kernel to the next
    step = ks[a:b]
    t0, t1 = step[0][0], ks[b][0]
    print(f"step wall {1e-3*(t1-t0):.1f} us, {len(step)} kernels", file=out)
    byq = collections.defaultdict(list)
    for k in step:
        byq[(k[3], k[4])].append(k)
    for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, *_ in v)
        print(f"queue {q}: {len(v)} kernels, busy {1e-3*busy:.1f} us, first +{1e-3*(v[0][0]-t0):.1f} us, last end +{1e-3*(v[-1][1]-t0):.1f} us", file=out)
    mainq = max(byq, key=lambda q: len(byq[q]))
    v = byq[mainq]
    others = [k for q, w in byq.items() if q != mainq for k in w]
    # main-queue gaps and overlap with the other queues
    gaps = collections.Counter()
    gap_total = 0
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for i, (s, e, n, *_q) in enumerate(v):
        ov = sum(max(0, min(e, oe) - max(s, os_)) for os_, oe, *_r in others)
        r = agg[short(n)]
        r[0] += 1
        r[1] += e - s
        r[2] += ov
        if i:
            g = s - v[i - 1][1]
            gap_total += max(g, 0)
            gaps[short(v[i - 1][2]) + " -> " + short(n)] += max(g, 0)
    print(f"main queue {mainq}: idle between kernels {1e-3*gap_total:.1f} us over {len(v)-1} boundaries", file=out)
    print("largest idle gaps on the main queue (summed per kernel pair):", file=out)
    for k, g in gaps.most_common(12):
        print(f"  {1e-3*g:8.1f} us  {k}", file=out)
    print("main-queue kernels: count, total us, of which overlapped by other queues", file=out)
    for k, (c, t, o) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {c:4d} {1e-3*t:9.1f} {1e-3*o:9.1f}  {k}", file=out)
    print("other queues:", file=out)
    agg2 = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, *_q in others:
        agg2[short(n)][0] += 1
        agg2[short(n)][1] += e - s
    for k, (c, t) in sorted(agg2.items(), key=lambda kv: -kv[1][1]):
        print(f"  {c:4d} {1e-3*t:9.1f}  {k}", file=out)
    print("main-queue sequence (start us, dur us, gap-before us, kernel):", file=out)
    for i, (s, e, n, *_q) in enumerate(v):
        g = s - v[i - 1][1] if i else 0
        print(f"  {1e-3*(s-t0):9.1f} {1e-3*(e-s):8.1f} {1e-3*g:7.1f}  {short(n)}", file=out)


if __name__ == "__main__":
    main()
