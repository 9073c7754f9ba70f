#!/bin/bash
# Builds variants of the library that differ in -D switches of traverse.hip (measurements only):
#   tools/build_variants.sh name1:"-DX=1 -DY=2" name2:"-DX=3" ...   ->  renderer_amd/variants/libinvert_<name>.so
# A name of the form base@<git-rev> takes traverse.hip from that revision instead.
set -e
cd "$(dirname "$0")/../renderer_amd/csrc"
make -s
mkdir -p ../variants
OTHER="kernels.o lbvh.o host_api.o obj_loader.o camera.o io_extras.o comm.o"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I. -I../../include"
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"; [ "$defs" = "$spec" ] && defs=""
  src=traverse.hip
  case "$name" in *@*) rev="${name#*@}"; name="${name%%@*}"; git show "$rev:renderer_amd/csrc/traverse.hip" > ../variants/_$name.hip; src=../variants/_$name.hip;; esac
  /opt/rocm/bin/hipcc $FLAGS $defs -c $src -o ../variants/$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/libinvert_$name.so ../variants/$name.o $OTHER -ldl
  echo built $name "($defs)"
done
