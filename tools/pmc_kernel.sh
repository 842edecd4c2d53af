#!/usr/bin/env bash
# usage: tools/pmc_kernel.sh <kernel-name-substring> <outdir> -- <python args...>
set -u
PAT=$1; out=$2; shift 3
export TMPDIR=/tmp
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "TCC_HIT TCC_MISS TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY" \
           "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES" \
           "FETCH_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$PAT" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob("$out/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$PAT" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open("$out/summary.txt", "w") as fh:
    if dur:
        line = f"kernel duration (pmc pass 1) n={len(dur)} avg_us={sum(dur)/len(dur)/1e3:.2f} min_us={min(dur)/1e3:.2f}"
        print(line); fh.write(line + "\n")
    for k in sorted(acc):
        v = acc[k]
        line = f"{k:32s} n={len(v):3d} avg={sum(v)/len(v):16.1f}"
        print(line); fh.write(line + "\n")
PY
