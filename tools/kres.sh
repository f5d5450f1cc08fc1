#!/bin/bash
# kernel resource usage (VGPRs / SGPRs / scratch / LDS) of one source file: tools/kres.sh k_stft.hip [extra flags]
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c audfprint_amd/csrc/$f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|TotalSGPRs|  VGPRs:|Scratch|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - 
