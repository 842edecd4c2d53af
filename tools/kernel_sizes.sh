#!/bin/bash
# code size (bytes) of every kernel in libf5hip.so; anything near 64 KB thrashes the instruction cache
set -e
T=$(mktemp -d); cd $T
objcopy -O binary --only-section=.hip_fatbin /root/repo/korean-f5-tts_amd/libf5hip.so fat.bin
python3 - <<'PY'
import re
d=open('fat.bin','rb').read()
offs=[m.start() for m in re.finditer(b'\x7fELF',d)]
for i,o in enumerate(offs):
    open(f'co{i}.elf','wb').write(d[o:(offs[i+1] if i+1<len(offs) else len(d))])
PY
for f in co*.elf; do /opt/rocm/lib/llvm/bin/llvm-readelf -sW $f 2>/dev/null | grep FUNC | awk '{print $3, $8}'; done | c++filt | awk '{s=$1; $1=""; print s, substr($0,1,160)}' | sort -n | uniq | tail -${1:-25}
rm -rf $T
