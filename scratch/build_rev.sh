#!/bin/bash
# build_rev.sh <name> <git rev | WORK>: libmoe355 built from the csrc/ and include/ of a revision (WORK = the working
# tree) -> scratch/exp/<name>.so, for same-box A/B runs with scratch/variants.sh.  Extra arguments go to hipcc.
set -e
name=$1; rev=$2; shift 2
root=$(cd $(dirname $0)/.. && pwd)
dst=$root/scratch/exp/rev_$name
rm -rf $dst; mkdir -p $dst/gpt-oss-vllm_amd/csrc $dst/include
if [ "$rev" = "WORK" ]; then
  cp $root/gpt-oss-vllm_amd/csrc/*.hip $root/gpt-oss-vllm_amd/csrc/*.h $root/gpt-oss-vllm_amd/csrc/Makefile $dst/gpt-oss-vllm_amd/csrc/
  cp $root/include/*.h $dst/include/
else
  for f in $(git -C $root ls-tree --name-only $rev gpt-oss-vllm_amd/csrc/ include/); do
    git -C $root show $rev:$f > $dst/$f
  done
fi
extra="$*"
make -s -C $dst/gpt-oss-vllm_amd/csrc -j8 LIB=$root/scratch/exp/$name.so CXXFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -I$dst/include -Wall -Wno-unused-function $extra"
echo built $root/scratch/exp/$name.so
