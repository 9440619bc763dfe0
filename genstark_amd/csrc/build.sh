#!/bin/bash
# Builds libgstark_hip.so for gfx950 (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
UNITS="ctx ntt pointwise hash air_mimc air_vm air_jit small"
HDRS="gf128_lazy.h host_pow.h gf128.h gf_small.h gf_wide.h common.h host_field.h host_field_small.h host_field_wide.h host_sha256.h hash_core.h ../../include/gstark.h"
# one library per field: the 128-bit field of the hot path, and two "plumbing" flavours of the same sources for the small prime
# fields of the reference's examples (gf_small.h): 2^64 - 21*2^30 + 1 (rescue/hash2x64.ts) and 2^32 - 3*2^25 + 1 (demo/fibonacci.ts)
# the field headers as string literals: the source text hiprtc compiles AIR programs against (air_jit.hip)
for pair in gf128.h:jit_gf128.inc gf128_lazy.h:jit_gf128_lazy.inc gf_small.h:jit_gf_small.inc gf_wide.h:jit_gf_wide.inc; do
  src=${pair%%:*}; dst=${pair##*:}
  if [ ! -f $dst ] || [ $src -nt $dst ]; then { printf 'R"GSJIT('; cat $src; printf ')GSJIT"\n'; } > $dst; fi
done
build_flavour() {   # <object dir> <output> <extra flags>
  local dir=$1 out=$2 extra=$3 pids=() stale=0
  mkdir -p $dir
  for f in $UNITS; do
    local need=0
    [ -f $dir/$f.o ] || need=1
    for h in $f.hip $HDRS; do [ $h -nt $dir/$f.o ] && need=1; done
    if [ $need = 1 ]; then $HIPCC $FLAGS $extra -c $f.hip -o $dir/$f.o & pids+=($!); stale=1; fi
  done
  for p in "${pids[@]}"; do wait $p; done
  for f in $UNITS; do [ $dir/$f.o -nt $out ] && stale=1; done      # an object compiled by hand is newer than the library too
  if [ $stale = 1 ] || [ ! -f $out ]; then
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o $out $(for f in $UNITS; do echo $dir/$f.o; done) -lhiprtc -ldl
  fi
  echo built $(pwd)/$out
}
# the compiler of AIR programs as a process of its own (jitc.cc): background builds of every flavour go through it
if [ ! -f gstark_jitc ] || [ jitc.cc -nt gstark_jitc ]; then
  g++ -O2 -std=c++17 -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include jitc.cc -o gstark_jitc -L/opt/rocm/lib -lhiprtc -Wl,-rpath,/opt/rocm/lib
fi
build_flavour build libgstark_hip.so "" &
build_flavour build_q64 libgstark_hip_q64.so "-DGS_SMALL_Q=18446744051160973313ull" &
build_flavour build_q32 libgstark_hip_q32.so "-DGS_SMALL_Q=4194304001ull" &
build_flavour build_q17 libgstark_hip_q17.so "-DGS_SMALL_Q=96769ull" &        # examples/demo/staticVariables.ts:11
# ... and the two multi-limb fields (gf_wide.h, 32-byte elements): 2^256 - 351*2^32 + 1 (mimc/mimc256.ts), 2^224 - 2^96 + 1 (lib224.aa)
build_flavour build_p256 libgstark_hip_p256.so "-DGS_WIDE_BITS=256" &
build_flavour build_p224 libgstark_hip_p224.so "-DGS_WIDE_BITS=224" &
# ... and the flavour whose modulus is given at run time (gs_set_modulus: any odd modulus below 2^256; generic kernels, two Montgomery
# reductions per product): index.ts:14's createPrimeField(modulus) for a modulus no fixed build knows
build_flavour build_rt libgstark_hip_rt.so "-DGS_WIDE_BITS=0" &
wait
# static VALU instruction mix of the NTT pass kernels (bench.py: roofline.second_roof): from the device assembly of ntt.hip
if [ ! -f ntt_isa_mix.json ] || [ ntt.hip -nt ntt_isa_mix.json ] || [ gf128_lazy.h -nt ntt_isa_mix.json ]; then
  $HIPCC $FLAGS -S --cuda-device-only ntt.hip -o build/ntt.s 2>/dev/null && python3 ../../tools/valu_mix.py build/ntt.s > ntt_isa_mix.json.tmp && mv ntt_isa_mix.json.tmp ntt_isa_mix.json
fi
# the native prove() driver: plain C++ above the C ABI (binds to whichever implementation of it the caller loaded).  One build per
# field flavour, like the ABI library: its host-side scalars (domain roots, Fiat-Shamir coefficients, interpolants, the remainder check)
# are computed in that flavour's host arithmetic, elements are gs_element_size() bytes
build_driver() {   # <output> <extra flags>
  local out=$1 extra=$2
  local stale=0 d
  # everything prover.cc includes, directly or through prover_dist.h / verifier.h (g++ -MM prover.cc lists the same files)
  for d in prover.cc prover_dist.h verifier.h host_sha256.h host_field.h host_field_small.h host_field_wide.h host_pow.h gf_wide.h \
           ../../include/gstark.h ../../include/gstark_comm.h ../../include/gstark_prover.h; do
    [ $d -nt $out ] && stale=1
  done
  if [ ! -f $out ] || [ $stale = 1 ]; then
    g++ -O3 -std=c++17 -shared -fPIC -Wall -Wno-unused-function -Wno-unknown-pragmas $extra prover.cc -ldl -o $out
  fi
}
build_driver libgstark_prover.so "" &
build_driver libgstark_prover_q64.so "-DGS_SMALL_Q=18446744051160973313ull" &
build_driver libgstark_prover_q32.so "-DGS_SMALL_Q=4194304001ull" &
build_driver libgstark_prover_q17.so "-DGS_SMALL_Q=96769ull" &
build_driver libgstark_prover_p256.so "-DGS_WIDE_BITS=256" &
build_driver libgstark_prover_p224.so "-DGS_WIDE_BITS=224" &
build_driver libgstark_prover_rt.so "-DGS_WIDE_BITS=0" &
wait
echo built $(pwd)/libgstark_prover.so
# the communicator of a distributed proof over RCCL / xGMI (include/gstark_comm.h): host code against librccl + the HIP runtime
if [ ! -f libgstark_rccl.so ] || [ comm_rccl.cc -nt libgstark_rccl.so ] || [ ../../include/gstark_comm.h -nt libgstark_rccl.so ]; then
  $HIPCC --offload-arch=gfx950 -O2 -std=c++17 -shared -fPIC -Wall comm_rccl.cc -o libgstark_rccl.so -L/opt/rocm/lib -lrccl -ldl
fi
echo built $(pwd)/libgstark_rccl.so
