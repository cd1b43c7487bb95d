#!/bin/bash
# Build an A/B variant of librecbox_hip.so:  variant.sh NAME SRC.hip -DMACRO=...   (run here, not on the GPU box)
# -> recbox_amd/lib/librecbox_hip_NAME.so ; select it with RECBOX_HIP_LIB=... on the GPU box.
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; shift 2
python -m recbox_amd.build >/dev/null
obj=/tmp/variant_${name}.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Irecbox_amd/csrc "$@" -c recbox_amd/csrc/$src -o $obj
objs=$(ls recbox_amd/build/*.o | grep -v "/${src%.hip}\.")
hipcc --offload-arch=gfx950 -shared -fPIC -o recbox_amd/lib/librecbox_hip_${name}.so $objs $obj
echo built recbox_amd/lib/librecbox_hip_${name}.so
