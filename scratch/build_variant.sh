#!/bin/bash
# build_variant.sh <name> "<extra hipcc flags>": libmoe355 with extra defines -> scratch/exp/<name>.so
set -e
name=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
out=$root/scratch/exp; mkdir -p $out/obj_$name
for f in api dequant router moe moe_tile attn attn_prefill ep layer dense dense_tile; do
  src=$root/gpt-oss-vllm_amd/csrc/$f.hip
  obj=$root/gpt-oss-vllm_amd/csrc/build/$f.o
  if [ "$f" = "moe_tile" ] || [ "$f" = "attn" ] || [ "$f" = "router" ] || [ "$f" = "attn_prefill" ] || [ "$f" = "moe" ] || [ "$f" = "dense_tile" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -I$root/include -Wall -Wno-unused-function "$@" -c $src -o $out/obj_$name/$f.o &
    objs="$objs $out/obj_$name/$f.o"
  else
    objs="$objs $obj"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $objs
echo built $out/$name.so
