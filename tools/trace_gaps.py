"""Kernel-boundary anatomy from a rocprofv3 --kernel-trace CSV: per kernel name the average duration and the average idle
gap between its end and the next kernel's start on the same queue (steady-state region: the last `frac` of the trace)."""
import csv, sys, collections
f = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    n = a["Kernel_Name"].replace("f5::", "")[:70]
    dur[n].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 200000:   # ignore the host-side pauses between utterances
        gap[n].append(g)
tot_d = sum(sum(v) for v in dur.values()); tot_g = sum(sum(v) for v in gap.values())
print(f"kernels {len(rows)}  busy {tot_d/1e6:.2f} ms  gaps {tot_g/1e6:.2f} ms ({100*tot_g/(tot_d+tot_g):.1f} % of the timeline)")
for n in sorted(dur, key=lambda k: -sum(dur[k]))[:14]:
    d = dur[n]; g = gap[n] or [0]
    print(f"{sum(d)/1e6:8.2f} ms  n={len(d):6d}  avg {sum(d)/len(d)/1e3:7.2f} us  gap after {sum(g)/len(g)/1e3:5.2f} us   {n}")
