#!/bin/sh
# TEST INFRASTRUCTURE: compile the HIP kernels for the host thread-per-lane emulator (tests/emu/hip_emu.hpp).
set -e
cd "$(dirname "$0")/../.."
exec g++ -std=c++20 ${EMU_CXXFLAGS:--O1} -g -fPIC -shared -DFRAD_HOST_EMULATION -Itests/emu -ffp-contract=off \
    -x c++ frad_python_amd/csrc/frad_hip.hip -o tests/emu/libfrad_emu.so -lpthread
