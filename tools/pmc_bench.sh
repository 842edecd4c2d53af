#!/usr/bin/env bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes as the guide prescribes) per kernel over one bench run.
set -u
export TMPDIR=/tmp
out=gpurun_out/pmc_bench
rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d $out/$n -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-precisions --no-c4 > $out/$n.log 2>&1 || echo "pass $n failed"
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_bench/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for a, b in (("f5::", ""), ("gemm_tn_glds_kernel", "G2"), ("gemm_tn_kernel", "G1")):
            k = k.replace(a, b)
        acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in acc.items():
    n = max(len(v) for v in d.values())
    fs = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [1])), 1)
    ws = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [1])), 1)
    rows.append((n * (fs + ws), k, n, fs, ws, d))
rows.sort(reverse=True)
with open("gpurun_out/pmc_bench/summary.txt", "w") as fh:
    hdr = "kernel | launches | FETCH_SIZE KB/launch (x2 for wide coalesced reads on gfx950) | WRITE_SIZE KB/launch | mfma_busy/launch | wave_cycles/launch | wait_any/launch | lds_conflict"
    print(hdr); fh.write(hdr + "\n")
    for tot, k, n, fs, ws, d in rows[:14]:
        g = lambda c: sum(d.get(c, [0])) / max(len(d.get(c, [1])), 1)
        line = f"{k} | {n} | {fs:.0f} | {ws:.0f} | {g('SQ_VALU_MFMA_BUSY_CYCLES'):.0f} | {g('SQ_WAVE_CYCLES'):.0f} | {g('SQ_WAIT_ANY'):.0f} | {g('SQ_LDS_BANK_CONFLICT'):.0f}"
        print(line); fh.write(line + "\n")
# HBM-side bytes per launch of the DiT-block GEMM kernels (launch-weighted), for bench.py's roofline.traffic
import json
g = [(k, n, fs, ws) for tot, k, n, fs, ws, d in rows if k.startswith("G2<") or "gemm_tn_glds" in k or "G2I" in k or "gemm_pp" in k]
g = [x for x in g if x[1] >= 300]          # the per-block kernels (>= 22 x 16 launches), not the once-per-utterance ones
if g:
    nl = sum(x[1] for x in g)
    fetch_kb = sum(x[1] * x[2] for x in g) / nl
    write_kb = sum(x[1] * x[3] for x in g) / nl
    tj = {"source": "tools/pmc_bench.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-precisions --no-c4`; launch-weighted over the DiT-block GEMM kernels",
          "collected": "rocprofv3 --pmc passes of this command, profiles/r02_pmc_per_kernel.txt",
          "kernels": [x[0] for x in g], "launches": nl,
          "fetch_kb_per_launch_raw": fetch_kb, "fetch_bytes_per_launch_corrected_x2": fetch_kb * 1024 * 2,
          "write_bytes_per_launch": write_kb * 1024, "traffic_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
          "note": "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads on gfx950; Infinity-Cache hits are counted by these fabric-side counters"}
    json.dump(tj, open("gpurun_out/pmc_bench/pmc_traffic.json", "w"), indent=1)
    print("traffic bytes/launch", tj["traffic_bytes_per_launch"])
PY
