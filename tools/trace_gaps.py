#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 --kernel-trace CSV: per stream (queue) the kernels' durations and the gaps between consecutive kernels;
how many kernels run at once on average.  usage: trace_gaps.py kernel_trace.csv [skip_first_n_kernels]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?"), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
ev.sort()
ev = ev[skip:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
print(f"{len(ev)} kernels over {(t1 - t0) / 1e6:.3f} ms; queues: {sorted(set(e[3] for e in ev))}")
byq = collections.defaultdict(list)
for e in ev: byq[e[4]].append(e)
for q, lst in sorted(byq.items()):
    lst.sort()
    busy = sum(e[1] - e[0] for e in lst)
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    print(f"  stream {q}: {len(lst)} kernels, busy {busy / 1e6:.3f} ms = {busy / (t1 - t0):.2f} of the span, mean gap {sum(gaps) / max(len(gaps), 1) / 1e3:.1f} us, gaps > 20 us: {sum(g > 20000 for g in gaps)}")
# concurrency
pts = sorted([(e[0], 1) for e in ev] + [(e[1], -1) for e in ev])
cur, last, acc = 0, t0, collections.Counter()
for t, d in pts:
    acc[cur] += t - last; last = t; cur += d
tot = sum(acc.values())
print("  kernels running at once: " + ", ".join(f"{k}: {v / tot:.2f}" for k, v in sorted(acc.items())))
byname = collections.defaultdict(list)
for e in ev: byname[e[2][:40]].append((e[1] - e[0]) / 1e3)
for n, v in sorted(byname.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:42s} n={len(v):4d} mean {sum(v) / len(v):7.1f} us  total {sum(v) / 1e3:7.3f} ms")
