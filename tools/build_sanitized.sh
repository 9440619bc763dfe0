#!/bin/bash
# tools/build_sanitized.sh [outdir] — AddressSanitizer + UndefinedBehaviorSanitizer builds of the host-side code that parses bytes it does
# not control: the native driver and verifier (csrc/prover.cc with prover_dist.h and verifier.h: serialized proofs, job structs) in the
# 128-bit and the 32-bit flavour, and the N-API addon (napi/gstark_napi.cc: lengths and types chosen by JavaScript).  CPU builds only
# (GPU sanitizers are not available on the pool).  tests/test_sanitizers.py drives them; tools/collect_round.sh reports the tier.
#   GSTARK_PROVER_LIB_DIR=<outdir>  makes genstark_amd/native.py load these drivers;   GSTARK_ADDON=<outdir>/gstark_napi.node  js/galois.js
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-/tmp/gstark_sanitized_$(id -u)}
mkdir -p "$out"
SAN="-O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
cd "$root/genstark_amd/csrc"
deps="prover.cc prover_dist.h verifier.h host_sha256.h host_field.h host_field_small.h host_field_wide.h host_pow.h gf_wide.h ../../include/gstark.h ../../include/gstark_comm.h ../../include/gstark_prover.h $root/tools/build_sanitized.sh"
build() {   # <output> <extra flags>
  local stale=0 d
  for d in $deps; do [ $d -nt "$out/$1" ] && stale=1; done
  if [ ! -f "$out/$1" ] || [ $stale = 1 ]; then
    g++ $SAN -std=c++17 -shared -fPIC -Wall -Wno-unused-function -Wno-unknown-pragmas $2 prover.cc -ldl -o "$out/$1.tmp.$$" && mv -f "$out/$1.tmp.$$" "$out/$1"
  fi
}
build libgstark_prover.so "" &
build libgstark_prover_q32.so "-DGS_SMALL_Q=4194304001ull" &
wait
if [ -f /usr/include/node/node_api.h ]; then
  cd "$root/napi"
  stale=0
  for d in gstark_napi.cc ../include/gstark.h ../include/gstark_prover.h "$root/tools/build_sanitized.sh"; do [ $d -nt "$out/gstark_napi.node" ] && stale=1; done
  if [ ! -f "$out/gstark_napi.node" ] || [ $stale = 1 ]; then
    g++ $SAN -std=c++17 -shared -fPIC -I/usr/include/node gstark_napi.cc -o "$out/gstark_napi.node.tmp.$$" -ldl && mv -f "$out/gstark_napi.node.tmp.$$" "$out/gstark_napi.node"
  fi
fi
echo "$out"
