#!/bin/bash
# Development build of libcvx_amd.so with clock stamps in k_he_lm (-DHE_LM_PROF): scripts/micro/libcvx_amd_heprof.so.
# Run scripts/handeye_prof.py against it with CVX_AMD_LIB=scripts/micro/libcvx_amd_heprof.so.
set -e
cd "$(dirname "$0")/../cvx_amd/csrc"
make -s
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wno-unused-function -DHE_LM_PROF -c handeye.hip -o build/handeye_prof.o
hipcc --offload-arch=gfx950 -shared -o ../../scripts/micro/libcvx_amd_heprof.so build/calib.o build/handeye_prof.o build/homography.o build/toolba.o build/lm_driver.o build/host_util.o build/handeye_init.o build/io_formats.o build/rccl_exchange.o -ldl
