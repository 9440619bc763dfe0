#!/bin/bash
# builds the N-API shim against the node headers of this image (no node-gyp, no network)
set -e
cd "$(dirname "$0")"
g++ -O2 -std=c++17 -shared -fPIC -I/usr/include/node gstark_napi.cc -o gstark_napi.node -ldl
echo built $(pwd)/gstark_napi.node
