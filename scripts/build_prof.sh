#!/bin/bash
# Development build of libcvx_amd.so with in-kernel cycle counters (-DK2_PROF): scripts/micro/libcvx_amd_prof.so.
# Run scripts/k2_prof.py against it with CVX_AMD_LIB=scripts/micro/libcvx_amd_prof.so.
set -e
cd "$(dirname "$0")/../cvx_amd/csrc"
make -s
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wno-unused-function -DK2_PROF -c calib.hip -o build/calib_prof.o
hipcc --offload-arch=gfx950 -shared -o ../../scripts/micro/libcvx_amd_prof.so build/calib_prof.o build/handeye.o build/homography.o build/toolba.o build/lm_driver.o build/host_util.o build/handeye_init.o build/io_formats.o build/rccl_exchange.o -ldl
