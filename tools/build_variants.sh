#!/bin/bash
# Build experiment variants of the library into ra_amd/csrc/variants/ (git-ignored, they travel with gpurun):
#   tools/build_variants.sh name1:"-DFLAG=1 -DOTHER=2" name2:"..."
# Every variant instantiates N=5 only (-DRGB_X_ONLY_N=5): bench.py --members 5 with RGB_LIB=<variant>.
# tools/ab_variants.sh times every variants/*.so with the same commands on the same box (interleaved, repeated).
set -u
cd "$(dirname "$0")/../ra_amd/csrc"
mkdir -p variants
pids=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-parameter -mllvm -disable-machine-licm -DRGB_X_ONLY_N=${ONLY_N:-5} $flags -shared \
      -o variants/$name.so rgb_kernels.hip rgb_api.hip rgb_wal.hip rgb_wal_host.cpp rgb_comm.cpp -ldl > variants/$name.log 2>&1 \
      && echo "built $name ($flags)" || { echo "FAILED $name"; tail -5 variants/$name.log; } ) &
  pids+=($!)
  if [ ${#pids[@]} -ge ${JOBS:-6} ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
