#!/bin/bash
# kernel resource usage (VGPR / scratch / occupancy / LDS) of one object: tools/kres.sh hc [filter]
cd "$(dirname "$0")/.."
src=e2-tts-pytorch_amd/csrc/$1.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I include -I e2-tts-pytorch_amd/csrc \
  -Rpass-analysis=kernel-resource-usage -c $src -o /dev/null 2>&1 | grep "remark:" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | \
  awk '/Function Name/{name=$3} /^VGPRs:/{v=$2} /^AGPRs:/{a=$2} /ScratchSize/{sc=$3} /Occupancy/{o=$3} /LDS Size/{print name, "vgpr="v, "agpr="a, "scratch="sc, "occ="o, "lds="$4}' | \
  c++filt | sed -e 's/(anonymous namespace):://g' -e 's/(HC[A-Za-z]*)//' | grep -E "${2:-.}"
