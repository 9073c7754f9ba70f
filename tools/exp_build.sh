#!/bin/bash
# A variant of the library for an experiment: lbvh.hip (or the file named third) compiled again with extra defines and
# linked with the objects of the regular build.
#     bash tools/exp_build.sh NAME "-DSOMETHING ..." [file.hip]    ->  renderer_amd/libinvert_hip_exp_NAME.so
set -e
cd "$(dirname "$0")/../renderer_amd/csrc"
name=$1; defs=$2; src=${3:-lbvh.hip}
make -s
mkdir -p exp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 $defs -c $src -o exp/$name.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libinvert_hip_exp_$name.so $objs exp/$name.o -ldl
echo built renderer_amd/libinvert_hip_exp_$name.so
