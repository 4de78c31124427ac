#!/bin/bash
# usage: [BN_SCHED="" for the default scheduler] tools/build_variant.sh <name> [extra hipcc flags...]  -> bn254_amd/ab/lib_<name>.so  (a library variant for same-box A/B runs: tools/ab.sh)
mkdir -p "$(dirname "$0")/../bn254_amd/ab"
name=$1; shift
hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wl,--no-undefined ${BN_SCHED--mllvm -amdgpu-sched-strategy=max-ilp} "$@" -o "$(dirname "$0")/../bn254_amd/ab/lib_$name.so" "$(dirname "$0")"/../bn254_amd/csrc/*.hip
