#!/bin/bash
# registers / LDS / occupancy of the kernels of one source file (device-side compile only, ~30 s): tools/kres.sh net_conv.hip [name pattern]
cd "$(dirname "$0")/../golds-rl-gym_amd" && mkdir -p /tmp/kres
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../include -I/opt/rocm/include --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -c csrc/$1 -o /tmp/kres/x.co 2> /tmp/kres/res.txt
grep -A12 "Function Name: .*${2:-}" /tmp/kres/res.txt | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - | sed 's/Function Name: _ZN3grl[0-9]*//' | cut -c1-200
