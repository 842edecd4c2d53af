"""Per-kernel summary of the passes of tools/pmc_pass.sh: average duration (kernel-trace pass), FETCH_SIZE / WRITE_SIZE per launch
(raw and with the gfx950 x2 correction of MI355X_MICROARCH.md for wide coalesced reads), SQ counters; and the HBM-side traffic per
launch of the DiT-block GEMM class next to its algorithmic bytes -> <out>/pmc_traffic.json (bench.py's roofline.traffic)."""
import collections
import csv
import glob
import json
import re
import sys

wl, out = sys.argv[1], sys.argv[2]


def _targs(s, i):
    """template-argument list of an Itanium-mangled name, for the subset this library uses (c++filt of this image does not know
    DF16b): returns (list of strings, index after the closing E)"""
    out = []
    while i < len(s) and s[i] != "E":
        if s.startswith("DF16b", i):
            out.append("bf16"); i += 5
        elif s.startswith("DF16_", i):
            out.append("f16"); i += 5
        elif s[i] == "f":
            out.append("float"); i += 1
        elif s[i] == "L":                      # literal: L<type><value>E
            j = s.index("E", i)
            out.append(s[i + 2:j].replace("n", "-")); i = j + 1
        elif s.startswith("NS_", i) or s[i].isdigit():
            nested = s.startswith("NS_", i)    # N S_ <name> [I <args> E] E : a name inside namespace f5
            if nested:
                i += 3
            m = re.match(r"\d+", s[i:])
            n = int(m.group()); i += len(m.group())
            name = s[i:i + n]; i += n
            if i < len(s) and s[i] == "I":
                sub, i = _targs(s, i + 1)
                name += "<" + ", ".join(sub) + ">"
            if nested:
                i += 1                         # the E that closes the nested name
            out.append(name)
        else:
            break
    return out, i + 1


def demangle(k):
    """rocprofv3 leaves some kernel names mangled in the counter CSVs"""
    m = re.match(r"_ZN2f5(\d+)", k)
    if not m:
        return k
    n = int(m.group(1))
    i = m.end()
    name = k[i:i + n]
    i += n
    if i < len(k) and k[i] == "I":
        try:
            args, _ = _targs(k, i + 1)
            return "f5::" + name + "<" + ", ".join(args) + ">"
        except Exception:
            return "f5::" + name
    return "f5::" + name


def short(k):
    k = demangle(k)
    for a, b in (("f5::", ""), ("gemm_tn_glds_kernel", "G2"), ("gemm_tn_kernel", "G1"), ("gemm_pp_kernel", "G3"), ("(anonymous namespace)::", "")):
        k = k.replace(a, b)
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*$", "", k)
    k = k.replace("__hip_bfloat16", "bf16").replace("_Float16", "f16")
    return k[:110]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))


def avg(v):
    return sum(v) / len(v) if v else 0.0


rows = []
for k, d in acc.items():
    n = max(len(v) for v in d.values())
    rows.append((n * (2 * avg(d.get("FETCH_SIZE", [])) + avg(d.get("WRITE_SIZE", []))), k, n, d))
rows.sort(reverse=True)
lines = ["# tools/pmc_pass.sh %s: rocprofv3 --pmc passes over ONE eager sample() (tools/pmc_one.py), all on one box; durations from the "
         "--kernel-trace pass of the same script" % wl,
         "kernel | launches | avg_us | FETCH_SIZE KB/launch raw | x2 corrected MB | WRITE_SIZE MB/launch | HBM-side GB/s (corrected) | "
         "mfma_busy cycles | wave_cycles(quad) | wait_any | wait_inst_any | active_inst_any | lds_conflict"]
for tot, k, n, d in rows[:16]:
    f, w = avg(d.get("FETCH_SIZE", [])), avg(d.get("WRITE_SIZE", []))
    us = avg(dur.get(k, [])) / 1e3
    bw = (2 * f + w) * 1024 / (us * 1e-6) / 1e9 if us > 0 else 0
    g = lambda c: avg(d.get(c, []))
    lines.append(f"{k} | {n} | {us:.2f} | {f:.0f} | {2 * f * 1024 / 1e6:.2f} | {w * 1024 / 1e6:.2f} | {bw:.0f} | {g('SQ_VALU_MFMA_BUSY_CYCLES'):.0f} | "
                 f"{g('SQ_WAVE_CYCLES'):.0f} | {g('SQ_WAIT_ANY'):.0f} | {g('SQ_WAIT_INST_ANY'):.0f} | {g('SQ_ACTIVE_INST_ANY'):.0f} | {g('SQ_LDS_BANK_CONFLICT'):.0f}")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

# ---- DiT-block GEMM class: the kernels launched once per block per step (>= 22 x NFE launches each)
M = 2048 if wl == "c2" else 32768
nfe = 16 if wl == "c2" else 4
D, F, e = 1024, 2048, 2          # 16-bit operands
alg = {"qkv": M * D * e + 3 * D * D * e + 3 * M * D * e, "out": M * D * e + D * D * e + 2 * M * D * 4,
       "ff1": M * D * e + F * D * e + M * F * e, "ff2": M * F * e + D * F * e + 2 * M * D * 4}
alg_avg = sum(alg.values()) / 4
g = [(k, n, avg(d.get("FETCH_SIZE", [])), avg(d.get("WRITE_SIZE", []))) for tot, k, n, d in rows
     if (k.startswith("G2<") or k.startswith("G3<")) and n >= 22 * nfe]
if g:
    nl = sum(x[1] for x in g)
    fetch_kb = sum(x[1] * x[2] for x in g) / nl
    write_kb = sum(x[1] * x[3] for x in g) / nl
    tj = {"source": "tools/pmc_pass.sh %s -> tools/pmc_summary.py: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over ONE eager "
                    "sample() (tools/pmc_one.py), same box, same build; launch-weighted over the DiT-block GEMM kernels" % wl,
          "collected": "round 3, rocprofv3 --pmc passes of `python tools/pmc_one.py %s`" % wl,
          "kernels": [x[0] for x in g], "launches": nl, "rows_per_launch": M,
          "fetch_kb_per_launch_raw": fetch_kb, "fetch_bytes_per_launch_corrected_x2": fetch_kb * 1024 * 2,
          "write_bytes_per_launch": write_kb * 1024, "traffic_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
          "algorithmic_bytes_per_launch": alg_avg, "algorithmic_bytes_per_kernel": alg,
          "note": "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads on gfx950; Infinity-Cache hits are counted "
                  "by these fabric-side counters"}
    json.dump(tj, open(out + "/pmc_traffic.json", "w"), indent=1)
    print("GEMM class traffic bytes/launch %.3e vs algorithmic %.3e" % (tj["traffic_bytes_per_launch"], alg_avg))
