#!/usr/bin/env bash
# Runs the GPU stages one after another on the gpurun box; a stage that times out or is killed stops the chain
# (never start another GPU step after a hang), an ordinary test failure does not.
# usage: tools/gpu_ci.sh stage1 stage2 ...   (stages: kernels sample smoke bench prof)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, timeout_s, command...
  local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/ci.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/ci.log
  tail -n 25 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -ge 128 ]; then echo "!!! $name hung or was killed: stopping"; exit $rc; fi
  return 0
}
for st in "$@"; do
  case $st in
    kernels) run kernels 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x --timeout 300 ;;
    kernels_all) run kernels 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --timeout 300 ;;
    sample) run sample 900 python -m pytest tests/test_sample_gpu.py -q -m gpu -s --timeout 600 ;;
    configs) run configs 1100 python -m pytest tests/test_configs_gpu.py -q -m gpu -s --timeout 900 ;;
    gpu_all) run gpu_all 1000 python -m pytest tests -q -m gpu -s --timeout 600 ;;
    smoke) run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) run bench 600 python bench.py --steps 5 --warmup 2 ;;
    bench_nocpu) run bench_nocpu 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ;;
    prof) run prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-precisions --no-c4 ;;
    dist) run dist 600 python -m pytest tests/test_dist_gpu.py -q -m gpu -s --timeout 300 ;;
    pmc_c2) run pmc_c2 1000 tools/pmc_pass.sh c2 gpurun_out/pmc_c2 ;;
    pmc_c3) run pmc_c3 1000 tools/pmc_pass.sh c3chunk gpurun_out/pmc_c3chunk ;;
    selflaunch) run selflaunch 400 python bench.py --gpus 2 --steps 3 --warmup 1 --rehearse-one-gpu --no-cpu-baseline --c4-utts 64 ;;
    *) echo "unknown stage $st"; exit 2 ;;
  esac
done
