#!/usr/bin/env python
"""tools/host_timeline.py DIR -- from `rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o tl -- python tools/host_link_probe.py ...`:
the LAST host-memory call of the process as a timeline: per direction the bytes, the busy time (union of the copies' intervals), the rate
while busy and the idle gaps; the kernels' busy time; the span from the first copy in to the last copy out."""
import csv, glob, os, sys
d = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)


copies = []
for r in rows("*memory_copy_trace.csv"):
    copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"], int(r.get("Size", 0) or 0)))
kernels = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows("*kernel_trace.csv")]
copies.sort()
if not copies:
    sys.exit("no copies traced")
# the last call: walk back from the end until a gap of more than 20 ms between consecutive events
ev = sorted([(s, e) for s, e, *_ in copies] + [(s, e) for s, e, _ in kernels])
t_end = ev[-1][1]
t0 = ev[-1][0]
for s, e in reversed(ev):
    if t0 - e > 20e6:
        break
    t0 = min(t0, s)
print(f"last call: {1e-6 * (t_end - t0):.2f} ms from its first to its last device event")


def union(iv):
    iv = sorted(iv)
    busy, cur_s, cur_e, gaps = 0, None, None, []
    for s, e in iv:
        if cur_s is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
    if cur_s is not None:
        busy += cur_e - cur_s
    return busy, gaps


for direction in sorted({c[2] for c in copies}):
    mine = [(s, e, n) for s, e, dd, n in copies if dd == direction and s >= t0]
    if not mine:
        continue
    busy, gaps = union([(s, e) for s, e, _ in mine])
    nbytes = sum(n for *_, n in mine)
    span = max(e for _, e, _ in mine) - min(s for s, _, _ in mine)
    big = sorted(gaps)[-5:]
    print(f"{direction:>28}: {len(mine):4d} copies, {nbytes / 1e9:.3f} GB, busy {busy * 1e-6:.2f} ms ({nbytes / max(busy, 1):.1f} GB/s while busy), "
          f"span {span * 1e-6:.2f} ms, from {1e-6 * (min(s for s, _, _ in mine) - t0):.2f} to {1e-6 * (max(e for _, e, _ in mine) - t0):.2f} ms, "
          f"largest gaps {[round(g * 1e-6, 2) for g in big]} ms")
km = [(s, e) for s, e, _ in kernels if s >= t0]
if km:
    busy, gaps = union(km)
    print(f"{'kernels':>28}: {len(km):4d} launches, busy {busy * 1e-6:.2f} ms, from {1e-6 * (min(s for s, _ in km) - t0):.2f} to {1e-6 * (max(e for _, e in km) - t0):.2f} ms")
