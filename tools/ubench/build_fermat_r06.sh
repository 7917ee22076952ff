#!/bin/bash
# Builds one executable per configuration of tools/ubench/fermat_r06.hip into _variants/ (git-ignored, travels with gpurun).
#   tools/ubench/build_fermat_r06.sh "<name>:<-D flags>" ...
set -e
cd "$(dirname "$0")/../.."
mkdir -p _variants
for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    (
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags tools/ubench/${SRC:-fermat_r06.hip} -Iinclude -Lgalois_amd -lgalois_amd \
            -Wl,-rpath,'$ORIGIN/../galois_amd' -Rpass-analysis=kernel-resource-usage -o _variants/fr6_$name 2> _variants/fr6_$name.log || { echo "BUILD FAILED $name"; tail -5 _variants/fr6_$name.log; }
        printf "%-28s %s\n" "$name" "$(grep -A9 "Function Name: .*${KERNEL:-fermat_a_kernelILb0}" _variants/fr6_$name.log | grep -E "VGPRs:|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ' ')"
    ) &
    while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
done
wait
