#!/bin/bash
# Builds an experimental variant of the library: tools/build_variant.sh <name> <extra hipcc flags...>
# -> build_variants/lib_<name>.so  (use with RAYTRACER_AMD_LIB=build_variants/lib_<name>.so)
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build_variants
C=raytracer_amd/csrc
id=$(python3 -m raytracer_amd.build --source-id)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-value -DRT_BUILD_ID="\"$id\"" "$@" -c $C/rt_device.hip -o build_variants/rt_device_$name.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/lib_$name.so build_variants/rt_device_$name.o $C/host_scene.o $C/accel_build.o $C/lightcam_build.o $C/rt_group.o || exit 1
rm -f build_variants/rt_device_$name.o
echo built build_variants/lib_$name.so
