#!/bin/bash
# Kernel-variant build of the library for A/B measurements: scripts/build_variant.sh NAME FILE.hip "-DMACRO=VALUE ..."
# -> sgnn_amd/lib/variants/libsgnn_hip_NAME.so (all other objects taken from the regular build); load it with SGNN_LIB=...
set -e
NAME=$1; FILE=$2; DEFS=$3
cd "$(dirname "$0")/../sgnn_amd/csrc"
make -s
mkdir -p ../lib/variants
OBJ=../lib/variants/${FILE%.hip}_$NAME.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function $DEFS -c $FILE -o $OBJ
OTHERS=$(ls ../lib/*.o | grep -v "/${FILE%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libsgnn_hip_$NAME.so $OBJ $OTHERS
echo built sgnn_amd/lib/variants/libsgnn_hip_$NAME.so
