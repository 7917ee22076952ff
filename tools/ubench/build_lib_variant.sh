#!/bin/bash
# Builds _variants/libgfa_<name>.so: the product library with ONE translation unit recompiled under extra -D flags (tuning builds,
# loaded through GALOIS_AMD_LIB).   tools/ubench/build_lib_variant.sh <unit, e.g. gfa_ntt_m32> "<name>:<-D flags>" ...
set -e
cd "$(dirname "$0")/../.."
unit=$1; shift
mkdir -p _variants
for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    (
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result $flags -c galois_amd/csrc/$unit.hip \
            -Rpass-analysis=kernel-resource-usage -o _variants/${unit}_$name.o 2> _variants/${unit}_$name.log || { echo "BUILD FAILED $name"; tail -5 _variants/${unit}_$name.log; exit 1; }
        objs=$(ls galois_amd/_obj/*.o | grep -v "/$unit.o")
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=galois_amd/csrc/exports.map -o _variants/libgfa_$name.so $objs _variants/${unit}_$name.o
        rm -f _variants/${unit}_$name.o
        printf "%-24s " "$name"; grep -A9 "Function Name: .*${KERNEL:-kernel}" _variants/${unit}_$name.log | grep -E "VGPRs:|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//; s/ScratchSize \[bytes\/lane\]/scratch/' | tr '\n' ' '; echo
    ) &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 0.5; done
done
wait
