#!/bin/bash
# tools/trace_bench_box.sh — the host-side MiMC recurrence experiments (tools/trace_bench.cpp) on THIS machine's cores, under the
# compilers / flags the library could be built with.  Output: stdout.
cd "$(dirname "$0")/.."
lscpu | grep -E "Model name|MHz" | head -3
for cc in "/opt/rocm/lib/llvm/bin/clang++ -O3" "/opt/rocm/lib/llvm/bin/clang++ -O3 -mtune=znver5" "/opt/rocm/lib/llvm/bin/clang++ -O3 -mtune=znver4" "/opt/rocm/lib/llvm/bin/clang++ -O3 -march=x86-64-v3" "/opt/rocm/lib/llvm/bin/clang++ -O3 -march=x86-64-v3 -mtune=znver5" "/opt/rocm/lib/llvm/bin/clang++ -O3 -march=native"; do
  $cc tools/trace_bench.cpp -o /tmp/trace_bench_x 2>/dev/null || { echo "$cc: build failed"; continue; }
  echo "== $cc"
  /tmp/trace_bench_x | tail -4
done
