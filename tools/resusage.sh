#!/bin/bash
# kernel resource usage of one HIP source (VGPRs, spills, occupancy, LDS) as one line per kernel.  usage: tools/resusage.sh file.hip [name filter]
cd "$(dirname "$0")/../cv-diffusion-model_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
       /TotalSGPRs:/ {sg=$(NF-1)} / VGPRs:/ {vg=$(NF-1)} /AGPRs:/ {ag=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {oc=$(NF-1)} /SGPRs Spill/ {ss=$(NF-1)} /VGPRs Spill/ {vs=$(NF-1)}
       /LDS Size/ {printf "%-110s vgpr %3s agpr %3s sgpr %3s spill v%s s%s scratch %s occ %s lds %s\n", name, vg, ag, sg, vs, ss, sc, oc, $(NF-1)}' |
  { if [ -n "$2" ]; then grep -- "$2"; else cat; fi; } | c++filt 2>/dev/null | sed 's/llie:://g; s/(llie::IrbxArgs, int, int)//'
