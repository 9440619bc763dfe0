#!/bin/bash
# builds the N-API shim against the node headers of this image (no node-gyp, no network).  Incremental, and the output appears
# atomically (temporary name + rename): tests that run side by side call this script while others already have the addon loaded
set -e
cd "$(dirname "$0")"
out=gstark_napi.node
stale=0
for d in gstark_napi.cc build.sh ../include/gstark.h ../include/gstark_prover.h; do [ $d -nt $out ] && stale=1; done
if [ ! -f $out ] || [ $stale = 1 ]; then
  tmp=$out.tmp.$$
  g++ -O2 -std=c++17 -shared -fPIC -I/usr/include/node gstark_napi.cc -o $tmp -ldl
  mv -f $tmp $out
fi
echo built $(pwd)/$out
