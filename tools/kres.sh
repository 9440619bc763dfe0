#!/bin/bash
# kernel resource table of one HIP unit (VGPRs, scratch, occupancy) from hipcc's remarks: tools/kres.sh ntt.hip [pattern] [extra flags]
cd "$(dirname "$0")/../genstark_amd/csrc"
unit=$1; pat=${2:-.}; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 -Wno-unused-value -Wno-unused-result \
  -Rpass-analysis=kernel-resource-usage "$@" -c $unit -o /tmp/kres_$$.o 2>&1 | \
  awk '/Function Name:/{name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
       / VGPRs:/{v=$0; sub(/.* VGPRs: /,"",v); sub(/ \[.*/,"",v)}
       /ScratchSize/{s=$0; sub(/.*ScratchSize \[bytes\/lane\]: /,"",s); sub(/ \[.*/,"",s)}
       /Occupancy/{o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)}
       /LDS Size/{l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); printf "%-70s vgpr %4s scratch %4s occ %2s lds %6s\n", name, v, s, o, l}' | grep -E "$pat"
rm -f /tmp/kres_$$.o
