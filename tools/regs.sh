#!/bin/bash
# tools/regs.sh FILE.hip [extra hipcc flags]: VGPRs / spill bytes / occupancy of every kernel in the file
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' \
 | awk '/Name:/{if(n)print n, v, a, s, o, l; n=$NF} /VGPRs:/{if($1=="VGPRs:")v="v="$2} /AGPRs:/{a="a="$2} /ScratchSize/{s="spill="$NF} /Occupancy/{o="occ="$NF} /LDS Size/{l="lds="$NF} END{print n, v, a, s, o, l}' \
 | while read n rest; do echo "$(echo $n | c++filt | sed 's/(ecrad::SpectralArgs)//; s/ecrad:://g; s/void //') $rest"; done
