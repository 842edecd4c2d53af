#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats summaries of the bench workloads (kernel_stats.csv only is kept: gpurun_out/ is size-limited).
# usage: tools/prof_round.sh <tag> <bench args...>      e.g. tools/prof_round.sh c2_f16x3 --precision f16x3
set -u
export TMPDIR=/tmp
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-precisions --no-c4 --no-c3 "$@" > "$root/gpurun_out/prof_$tag.log" 2>&1
rc=$?
cd "$root"
f=$(ls $out/*/*kernel_stats.csv 2>/dev/null | head -n 1)
if [ -n "$f" ]; then cp "$f" gpurun_out/kernel_stats_$tag.csv; fi
rm -rf "$out"
echo "prof $tag rc=$rc"
[ $rc -eq 124 ] || [ $rc -ge 128 ] && exit $rc
exit 0
