#!/usr/bin/env bash
# samples socket power and the shader / memory clocks (rocm-smi, read-only) every 0.5 s while a command runs
# usage: tools/power_watch.sh <out.log> <command...>
out=$1; shift
( while true; do rocm-smi -d 0 --showpower --showclocks --csv 2>/dev/null | grep "^card" >> "$out"; sleep 0.5; done ) &
wp=$!
"$@"
rc=$?
kill $wp 2>/dev/null
exit $rc
