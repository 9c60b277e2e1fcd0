#!/bin/bash
# Experiment tooling: builds variants of libfora_hip.so under gpurun_out/variants/ -- one per argument "name:-DFLAG1,-DFLAG2".
# gpurun_out/ is git-ignored but not in .gpurunignore? it is excluded from the snapshot -> build into build_variants/ instead.
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"; [ "$flags" = "$spec" ] && flags=""
  flags="${flags//,/ }"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $flags -shared \
      -o "variants/lib_${name}.so" fora_amd/csrc/fora_hip.hip &
done
wait
ls -la variants/
