#!/bin/bash
# usage: build_variant.sh NAME "extra flags"   -> /root/repo/polar-codes_amd/libpolar_hip_NAME.so (only the (1024, L=8, fp32) register list kernel)
set -e
cd /root/repo/polar-codes_amd/csrc
mkdir -p build/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -ffp-contract=off -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__ -DPOLAR_SCL_DEV_ONE $2 -Rpass-analysis=kernel-resource-usage -c kernels_scl_reg.hip -o build/var/scl_reg_$1.o 2>&1 | grep "VGPRs Spill\|ScratchSize" | head -2
OBJS=$(ls build/*.o | grep -v kernels_scl_reg.o | grep -v _prof.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpolar_hip_$1.so $OBJS build/var/scl_reg_$1.o -lhiprtc -ldl
echo built $1
