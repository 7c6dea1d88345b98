#!/bin/bash
# Builds the library with the device assembly passed through tools/patch_isa.py between compiler and assembler.
# usage: tools/build_patched.sh <name> [extra hipcc flags...]   -> build_variants/lib_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build_variants
C=raytracer_amd/csrc
L=/opt/rocm/lib/llvm/bin
W=build_variants/patched_$name
mkdir -p $W
id=$(python3 -m raytracer_amd.build --source-id)
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-value -DRT_BUILD_ID=\"$id\" $*"
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S $C/rt_device.hip -o $W/dev.s
python3 tools/patch_isa.py $W/dev.s $W/dev_patched.s
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev_patched.s -o $W/dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.co $W/dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.co -output=$W/dev.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c $C/rt_device.hip -o $W/rt_device.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/lib_$name.so $W/rt_device.o $C/host_scene.o $C/accel_build.o $C/lightcam_build.o $C/rt_group.o
echo built build_variants/lib_$name.so
