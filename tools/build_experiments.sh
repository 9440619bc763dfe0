#!/bin/bash
# The experiments build of the 128-bit library: the same sources with -DGS_NTT_EXPERIMENTS, which compiles in the NTT variants that
# were measured and not adopted (matrix-core passes: tools/ntt_mfma.h; workgroup and canonical-limb kernels as A/B partners) and the
# environment switches that select them per call (GSTARK_NTT_MFMA / GSTARK_NTT_WAVE / GSTARK_NTT_LAZY / GSTARK_NTT_TWIDDLE_LOG).
# Output: tools/ab/libgstark_hip_exp.so (git-ignored).  The product library (genstark_amd/csrc/build.sh) has none of this.
set -e
cd "$(dirname "$0")/../genstark_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 -Wno-unused-function -Wno-unused-value -Wno-unused-result -DGS_NTT_EXPERIMENTS"
mkdir -p build_exp ../../tools/ab
pids=()
for f in ctx ntt pointwise hash air_mimc air_vm air_jit small; do
  stale=0
  for h in $f.hip ../../tools/ntt_mfma.h *.h ../../include/gstark.h; do [ $h -nt build_exp/$f.o ] && stale=1; done      # any header: the context struct is shared
  if [ ! -f build_exp/$f.o ] || [ $stale = 1 ]; then $HIPCC $FLAGS -c $f.hip -o build_exp/$f.o & pids+=($!); fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libgstark_hip_exp.so build_exp/*.o -lhiprtc
echo built tools/ab/libgstark_hip_exp.so
