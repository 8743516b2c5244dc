#!/bin/bash
# usage: tools/variant_run.sh <variant.so> <cmd...>: runs cmd with the library swapped for a build variant
set -e
cp rust-compression_amd/libbz2_mi355x.so /tmp/lib_default.so
cp "$1" rust-compression_amd/libbz2_mi355x.so; shift
"$@" || true
cp /tmp/lib_default.so rust-compression_amd/libbz2_mi355x.so
