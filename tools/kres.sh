#!/bin/bash
# kernel resource usage (VGPRs / spills / occupancy) of one HIP source: tools/kres.sh dist_amd/csrc/gemm_nt.hip [filter]
f=$1; pat=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
 sed 's/ \[-Rpass.*\]//' | awk '/Function Name:/{name=$NF} / VGPRs:/{v=$NF} /AGPRs:/{a=$NF} /VGPRs Spill/{sp=$NF} /Occupancy/{o=$NF} /LDS Size/{print name, "vgpr="v, "agpr="a, "spill="sp, "occ="o}' | c++filt | grep -E "$pat"
