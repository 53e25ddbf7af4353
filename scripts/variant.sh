#!/bin/bash
# builds a library variant for A/B runs: scripts/variant.sh <name> "<extra hipcc flags>"  ->  build/<name>/libpbrhip.so
# (used as PBRHIP_LIB=build/<name>/libpbrhip.so; build/ is git-ignored but travels to the GPU box)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build/$name
make -s -C $root/pbrlab_amd/csrc -j4 OBJ=$root/build/$name/ OUT=$root/build/$name/libpbrhip.so EXTRA="$*"
echo $root/build/$name/libpbrhip.so
