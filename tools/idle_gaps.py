"""Idle time of the device in a rocprofv3 --kernel-trace CSV: union of the kernels' [start, end) intervals against the span they cover,
and the largest gaps with the kernels either side.  usage: idle_gaps.py <kernel_trace.csv> [first_fraction last_fraction | steps i j]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
def wgs(r):
    try:
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        w = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        return g // max(w, 1)
    except (KeyError, ValueError):
        return 1 << 30


iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
big = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if wgs(r) >= 256)      # launches that can fill the 256 CUs
if len(sys.argv) > 3 and sys.argv[2] == "steps":          # window = train steps i .. j-1, delimited by the end of each step's last sgd_kernel
    ends = [e for _, e, n in iv if "sgd_kernel" in n][1::2]
    a, b = ends[int(sys.argv[3]) - 1], ends[int(sys.argv[4]) - 1]
    print(f"{int(sys.argv[4]) - int(sys.argv[3])} steps: {(b - a) / 1e6 / (int(sys.argv[4]) - int(sys.argv[3])):.3f} ms per step")
else:
    lo_f, hi_f = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
    t0, t1 = iv[0][0], max(e for _, e, _ in iv)
    a, b = t0 + (t1 - t0) * lo_f, t0 + (t1 - t0) * hi_f
iv = [x for x in iv if x[0] >= a and x[1] <= b]
busy, gaps = 0, []
cur_s, cur_e, last = iv[0][0], iv[0][1], iv[0][2]
for s, e, name in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last, name))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last = name
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
print(f"span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {(span - busy) / 1e6:.3f} ms ({100 * (span - busy) / span:.1f} %)  kernels {len(iv)}")
gaps.sort(reverse=True)
tot = {}
for g, p, n in gaps:
    key = (p[:60], n[:60])
    tot[key] = tot.get(key, 0) + g
for (p, n), g in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{g / 1e3:9.1f} us total  after {p:60s} before {n}")

# time during which no launch of >= 256 workgroups is running: the chip is at best partly filled
bw = [x for x in big if x[0] >= a and x[1] <= b]
cover, cs, ce = 0, bw[0][0], bw[0][1]
for s_, e_ in bw[1:]:
    if s_ > ce:
        cover += ce - cs
        cs, ce = s_, e_
    else:
        ce = max(ce, e_)
cover += ce - cs
print(f"no launch with >= 256 workgroups running: {(span - cover) / 1e6:.3f} ms of the span ({100 * (span - cover) / span:.1f} %)")
