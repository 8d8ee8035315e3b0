#!/bin/bash
# Build A/B variants of libtheseus_hip.so with extra -D flags into scratch/variants/<name>.so
# usage: tools/variants.sh name1:"-DFLAG1 -DFLAG2" name2:"..."
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
OUT=${THX_VARIANT_DIR:-$ROOT/theseus_amd/lib/variants}; mkdir -p $OUT   # (theseus_amd/lib travels to the GPU box; scratch/ does not)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  for f in pg_kernels chol_kernels vjp_kernels block_kernels pg2_kernels ba_kernels vjp2_kernels pgso3_kernels ba_vjp_kernels vjpso3_kernels vjp_unroll_kernels vjp_unroll3_kernels vjp_unroll_ba_kernels; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed $flags -c $ROOT/theseus_amd/csrc/$f.hip -o $OUT/${name}_$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/$name.so $OUT/${name}_pg_kernels.o $OUT/${name}_chol_kernels.o $OUT/${name}_vjp_kernels.o $OUT/${name}_block_kernels.o $OUT/${name}_pg2_kernels.o $OUT/${name}_ba_kernels.o $OUT/${name}_vjp2_kernels.o $OUT/${name}_pgso3_kernels.o $OUT/${name}_ba_vjp_kernels.o $OUT/${name}_vjpso3_kernels.o $OUT/${name}_vjp_unroll_kernels.o $OUT/${name}_vjp_unroll3_kernels.o $OUT/${name}_vjp_unroll_ba_kernels.o
  rm -f $OUT/${name}_*.o
  echo built $OUT/$name.so
done
