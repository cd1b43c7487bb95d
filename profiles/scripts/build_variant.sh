#!/bin/bash
# build_variant.sh NAME -DFLAG=... : a librecbox_hip.so built with extra macros, under recbox_amd/lib/variants/NAME.so
# (run a bench against it with RECBOX_HIP_LIB=recbox_amd/lib/variants/NAME.so)
set -e
name=$1; shift
root=$(cd $(dirname $0)/../.. && pwd)
mkdir -p $root/recbox_amd/lib/variants /tmp/variant_$name
objs=""
for src in $root/recbox_amd/csrc/*.hip; do
  o=/tmp/variant_$name/$(basename $src).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include -I$root/recbox_amd/csrc -Wno-unused-function "$@" -c $src -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/recbox_amd/lib/variants/$name.so $objs
echo built $name
