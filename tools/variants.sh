#!/bin/bash
# Build A/B variants of libtheseus_hip.so with extra -D flags into theseus_amd/lib/variants/<name>.so (THESEUS_HIP_LIB=<that> selects it)
# usage: tools/variants.sh name1:"-DFLAG1 -DFLAG2" name2:"..."       (THX_VARIANT_FILES="chol_kernels ba_kernels": the sources the
#        flags touch, default chol_kernels; everything else is linked from the in-tree build's objects, theseus_amd/lib/*.o)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
OUT=${THX_VARIANT_DIR:-$ROOT/theseus_amd/lib/variants}; mkdir -p $OUT   # (theseus_amd/lib travels to the GPU box; scratch/ does not)
FILES=${THX_VARIANT_FILES:-chol_kernels}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  objs=""
  for src in $ROOT/theseus_amd/csrc/*.hip; do
    f=$(basename $src .hip)
    if [[ " $FILES " == *" $f "* ]]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed $flags -c $src -o $OUT/${name}_$f.o &
      objs="$objs $OUT/${name}_$f.o"
    else
      [ -f $ROOT/theseus_amd/lib/$f.o ] || { echo "missing $ROOT/theseus_amd/lib/$f.o: run python __graft_entry__.py first"; exit 1; }
      objs="$objs $ROOT/theseus_amd/lib/$f.o"
    fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/$name.so $objs
  rm -f $OUT/${name}_*.o
  echo built $OUT/$name.so
done
