#!/bin/bash
# Builds libgstark_hip.so for gfx950 (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
mkdir -p build
pids=()
for f in ctx ntt pointwise hash air_mimc air_vm small; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ gf128.cuh -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ host_field.h -nt build/$f.o ] || [ host_sha256.h -nt build/$f.o ] || [ hash_core.cuh -nt build/$f.o ] || [ ../../include/gstark.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libgstark_hip.so build/ctx.o build/ntt.o build/pointwise.o build/hash.o build/air_mimc.o build/air_vm.o build/small.o
echo built $(pwd)/libgstark_hip.so
# the native prove() driver: plain C++ above the C ABI (binds to whichever implementation of it the caller loaded)
g++ -O2 -std=c++17 -shared -fPIC -Wall -Wno-unused-function prover.cc -ldl -o libgstark_prover.so
echo built $(pwd)/libgstark_prover.so
