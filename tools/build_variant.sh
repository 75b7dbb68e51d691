#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  -> build/libfovpt_<name>.so (A/B experiments)
set -e
NAME=$1; shift
cd /root/repo/fovpathtracing_optixcodelatest_amd/csrc
mkdir -p /tmp/v_$NAME /root/repo/build
for f in fovpt_api wavefront bvh_build; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -std=c++17 "$@" -c $f.hip -o /tmp/v_$NAME/$f.o &
done
hipcc -O2 -ffp-contract=off -fPIC -std=c++17 -c model_loader.cpp -o /tmp/v_$NAME/model_loader.o &
wait -n; wait -n; wait -n; wait -n
hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build/libfovpt_$NAME.so /tmp/v_$NAME/*.o -lz
