#!/bin/bash
# Development build for the rounding-source experiment: K2's accumulations, the 6x6 factor and the
# substitutions without fused multiply-adds -> scripts/micro/libcvx_amd_nc.so.
# Run scripts/parity_blocks.py against it with CVX_AMD_LIB=scripts/micro/libcvx_amd_nc.so.
set -e
cd "$(dirname "$0")/../cvx_amd/csrc"
make -s
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wno-unused-function -DCVX_NO_CONTRACT -c calib.hip -o build/calib_nc.o
hipcc --offload-arch=gfx950 -shared -o ../../scripts/micro/libcvx_amd_nc.so build/calib_nc.o build/handeye.o build/homography.o build/toolba.o build/lm_driver.o build/host_util.o build/handeye_init.o build/io_formats.o build/rccl_exchange.o -ldl
