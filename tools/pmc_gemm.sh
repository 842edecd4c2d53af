#!/usr/bin/env bash
# usage: tools/pmc_gemm.sh M N K CFG  -> per-counter averages for the GEMM kernel in gpurun_out/pmc_<cfg>.txt
set -u
M=$1; N=$2; K=$3; CFG=$4
export TMPDIR=/tmp
out=gpurun_out/pmc_${M}_${N}_${K}_${CFG}
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT TCC_MISS TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY" \
           "TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCR_TCP_STALL_CYCLES TA_TA_BUSY" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python tools/gemm_one.py $M $N $K $CFG > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_tn" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$out/summary.txt", "w") as fh:
    for k in sorted(acc):
        v = acc[k]
        line = f"{k:32s} n={len(v):3d} avg={sum(v)/len(v):16.1f}"
        print(line); fh.write(line + "\n")
PY
