#!/bin/bash
# Library variant for A/B runs: tools/micro/build_variant.sh <name> "<-D flags>" <file.hip>...   ->  tools/micro/build/lib_<name>.so
# (the named translation units recompiled with the flags, everything else taken from the regular build)
set -e
name=$1; defs=$2; shift 2
C=hisstools_library_amd/csrc
make -s -j8 -C $C
V=tools/micro/build/obj_$name; mkdir -p $V
objs=""
for o in $C/build/*.o; do
  b=$(basename $o .o); use=$o
  for f in "$@"; do
    if [ "$(basename $f .hip)" == "$b" ]; then
      extra=""; [ "$b" == "hcv_mac_tiled" ] && extra="-fno-slp-vectorize"
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $extra $defs -c $C/$b.hip -o $V/$b.o
      use=$V/$b.o
    fi
  done
  objs="$objs $use"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/build/lib_$name.so $objs -ldl
echo "built tools/micro/build/lib_$name.so"
