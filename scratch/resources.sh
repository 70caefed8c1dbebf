#!/bin/bash
# VGPRs / spills / scratch of every kernel of one source file.  usage: resources.sh moe.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I/root/repo/include -c /root/repo/gpt-oss-vllm_amd/csrc/$f -o /tmp/res_chk.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
python3 -c '
import sys,re
name=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: name=m.group(1); d={}
    for k in (" VGPRs:"," AGPRs:","SGPRs Spill:","VGPRs Spill:","ScratchSize [bytes/lane]:","Occupancy [waves/SIMD]:"):
        m=re.search(re.escape(k)+r" (\d+)",l)
        if m and name: d[k.strip(" :").split(" [")[0]]=m.group(1)
    if "LDS Size" in l and name:
        short=re.sub(r"EvPK.*","",name)
        print(short, d); name=None
'
