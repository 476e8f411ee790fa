#!/bin/bash
# dev: build a variant of ONE source file into .ab/<name>/libprecondition_amd.so (linked with the
# other in-tree objects).  .ab/ does not travel with gpurun: run this on the GPU box.
# usage: tools/ab_build.sh <name> <source.hip> [extra hipcc flags...]
set -e
name=$1; src=$2; shift 2
cs=precondition_amd/csrc
mkdir -p .ab/$name
base=$(basename ${src%.hip})
obj=.ab/$name/$base.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off "$@" -c $cs/$src -o $obj
others=$(ls $cs/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $obj -ldl -o .ab/$name/libprecondition_amd.so
echo "built .ab/$name/libprecondition_amd.so ($*)"
