#!/bin/bash
# dev: run a script against the current library and against gpurun_out/lib_before.so
cd $GRAFT_REPO_ROOT
cp precondition_amd/libprecondition_amd.so /tmp/lib_after.so
for tag in after before after before; do
  cp /tmp/lib_$tag.so precondition_amd/libprecondition_amd.so 2>/dev/null || cp .ab/lib_$tag.so precondition_amd/libprecondition_amd.so
  echo "== $tag"; timeout 300 python -u "$@" 2>&1 | grep -v "Warn\|amdgpu.ids"
done
cp /tmp/lib_after.so precondition_amd/libprecondition_amd.so
