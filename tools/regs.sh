#!/bin/bash
# usage: tools/regs.sh <file.hip> [kernel-name-substring]  -- register / spill usage per kernel as the compiler reports it
f=${1:-transform.hip}; pat=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form ${EXTRA} \
  -Rpass-analysis=kernel-resource-usage -c evcont_amd/csrc/$f -o /tmp/regs_$$.o 2>&1 |
  grep "remark:" | sed 's/.*remark: *//; s/ *\[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/^Function Name/{if (line) print line; line=$3; next} /^(TotalSGPRs|VGPRs|AGPRs|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill)/{line=line " | " $0} END{print line}' |
  c++filt | grep -E "$pat"
rm -f /tmp/regs_$$.o
