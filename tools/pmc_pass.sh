#!/usr/bin/env bash
# usage: tools/pmc_pass.sh <c2|c3chunk> <outdir>
# Separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share one, MI355X_MICROARCH.md "rocprofv3 PMC slots") over ONE
# eager sample() (tools/pmc_one.py: phase markers on stderr, no HIP graph), plus one --kernel-trace --stats pass for the durations;
# the program comes directly after `--`.  A pass that fails or times out ends the script: nothing else is started on the GPU.
set -u
wl=$1; out=$2
export TMPDIR=/tmp
rm -rf "$out"; mkdir -p "$out"
pass() {  # name, rocprofv3 options...
  local n=$1; shift
  echo "== pass $n: $(date +%T)" | tee -a "$out/passes.log"
  timeout -k 10 240 rocprofv3 "$@" --output-format csv -d "$out/$n" -- python tools/pmc_one.py "$wl" > "$out/$n.log" 2>&1
  local rc=$?
  grep "pmc_one" "$out/$n.log" | tail -3 | tee -a "$out/passes.log"
  echo "== pass $n rc=$rc" | tee -a "$out/passes.log"
  return $rc
}
pass trace --kernel-trace --stats &&
pass FETCH_SIZE --pmc FETCH_SIZE &&
pass WRITE_SIZE --pmc WRITE_SIZE &&
pass SQ --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE &&
python tools/pmc_summary.py "$wl" "$out"
