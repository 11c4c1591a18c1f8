#!/bin/bash
# build_variant.sh NAME "-DFLAG=1 ..." file.hip [file2.hip ...]: a whole-library variant with the given files recompiled with extra flags
# -> tools/_variants/lib_NAME.so (objects of the other sources come from the last regular build)
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/ivos-w_amd/csrc
mkdir -p $root/tools/_variants /tmp/variants_$name
objs=""
for o in $csrc/build/*.o; do
  b=$(basename $o .o)
  skip=0
  for f in "$@"; do [ "$b" = "$f" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -I$root/include -c $csrc/$f -o /tmp/variants_$name/$f.o
  objs="$objs /tmp/variants_$name/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/_variants/lib_$name.so $objs
ls -la $root/tools/_variants/lib_$name.so
