#!/bin/bash
# usage: build_variant.sh FILE NAME "extra flags"  -> polar-codes_amd/libpolar_hip_NAME.so: the in-tree objects with FILE.hip recompiled
# under the extra flags (e.g. kernels_scl_wide wstat "-DPOLAR_WIDE_DEV_ONE -DPOLAR_WIDE_STATS")
set -e
cd /root/repo/polar-codes_amd/csrc
mkdir -p build/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -ffp-contract=off -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__ $3 -Rpass-analysis=kernel-resource-usage -c $1.hip -o build/var/$1_$2.o 2>&1 | grep "Function Name\|VGPRs\|ScratchSize\|error" | grep -v AGPR | head -24
OBJS=$(ls build/*.o | grep -v "$1.o" | grep -v _prof.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpolar_hip_$2.so $OBJS build/var/$1_$2.o -lhiprtc -ldl
echo built $2
