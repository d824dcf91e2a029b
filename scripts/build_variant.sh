#!/bin/bash
# Development build of the library with extra compiler flags for calib.hip -> scripts/micro/libcvx_amd_<tag>.so
# (select it with CVX_AMD_LIB).  Usage: scripts/build_variant.sh <tag> <flags...>
set -e
tag=$1; shift
cd "$(dirname "$0")/../cvx_amd/csrc"
make -s
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wall -Wno-unused-function "$@" -c calib.hip -o build/calib_$tag.o
hipcc --offload-arch=gfx950 -shared -o ../../scripts/micro/libcvx_amd_$tag.so build/calib_$tag.o build/handeye.o build/homography.o build/toolba.o build/lm_driver.o build/host_util.o build/handeye_init.o build/io_formats.o build/rccl_exchange.o -ldl -pthread
