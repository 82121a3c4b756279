#!/bin/bash
# Diagnostic builds: recompile SOME translation units with extra flags and link them with the product objects into
# frad_python_amd/csrc/libfrad_hip_<name>.so (git-ignored; select it with FRAD_PROBE_LIB / STAMPS_LIB / KB_LIB in the tools).
#   tools/build_variant.sh stamps "-DFRAD_WAVE_STAMPS" frad_p0_wave frad_p1_wave
set -e
NAME=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../frad_python_amd/csrc"
make -j8 >/dev/null
mkdir -p build_st
OBJS=""
for u in $(sed -n 's/^UNITS *:= *//p' Makefile); do
    if [[ " $* " == *" $u "* ]]; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $FLAGS -c $u.hip -o build_st/${NAME}_$u.o &
        OBJS="$OBJS build_st/${NAME}_$u.o"
    else
        OBJS="$OBJS build/$u.o"
    fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libfrad_hip_${NAME}.so $OBJS
ls -la libfrad_hip_${NAME}.so
