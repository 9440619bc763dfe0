#!/bin/bash
# tools/ntt_noformat.sh — what the 16-byte <-> five-limb conversions cost in the 2^24-point transform: the product library against a
# build whose pass kernels skip them (-DGS_EXP_NO_FORMAT: register moves instead of lz_unpack / lz_pack; results are garbage, timings
# are the point).  A plan with two passes instead of three would save ONE THIRD of the difference.  Run on the GPU box.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/genstark_amd/csrc
mkdir -p build_nofmt $root/tools/ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 -Wno-unused-function -Wno-unused-value -Wno-unused-result -DGS_EXP_NO_FORMAT"
pids=()
for f in ctx ntt pointwise hash air_mimc air_vm air_jit small; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o build_nofmt/$f.o & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/ab/libgstark_hip_nofmt.so build_nofmt/*.o -lhiprtc -ldl
cd /tmp && export TMPDIR=/tmp
for lib in product nofmt; do
  rm -rf /tmp/nf_$lib
  extra=""
  if [ $lib = nofmt ]; then extra=$root/tools/ab/libgstark_hip_nofmt.so; fi
  rocprofv3 --kernel-trace --stats -d /tmp/nf_$lib -o t -- python3 $root/tools/ntt_only.py 24 $extra > /dev/null 2>&1
  echo "== $lib"
  python3 $root/tools/rocprof_summary.py $(find /tmp/nf_$lib -name t_results.db | head -1) | grep -E "k_ntt_wave|total kernel"
done
