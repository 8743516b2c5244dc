#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>
# Builds the library with extra flags into rust-compression_amd/build/var/<name>.so (git-ignored, but it
# travels to the GPU box); run it there with tools/variant_run.sh.
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/rust-compression_amd/csrc
O=$R/rust-compression_amd/build/var_$NAME
mkdir -p $O $R/rust-compression_amd/build/var
pids=()
for s in k_rle1 k_bwt k_mtf k_huff k_emit k_dec k_deflate engine dec_engine deflate_engine capi; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__ "$@" -c $C/$s.hip -o $O/$s.o &
  pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait $p || fail=1; done
if [ $fail -ne 0 ]; then echo "build_variant: a compile failed" >&2; rm -rf $O; exit 1; fi
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/rust-compression_amd/build/var/$NAME.so $O/*.o
rm -rf $O
ls -la $R/rust-compression_amd/build/var/$NAME.so
