#!/bin/bash
# usage: res.sh <pattern> [extra hipcc flags]  -- resource use of kernels matching pattern
pat=$1; shift
cd /tmp/res
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed --save-temps "$@" -c -o k.o /root/repo/screencounter_amd/csrc/scg_kernels.hip 2>&1 | grep -E "error" -A5
python3 - "$pat" <<'PY'
import re,sys
s=open('/tmp/res/scg_kernels-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size', s, re.S):
    blk=m.group(0)
    g=lambda k: re.search(r'\.%s:\s+(\S+)'%k, blk).group(1)
    name=g('name')
    if re.search(sys.argv[1], name):
        print(name[:100], 'vgpr',g('vgpr_count'),'sgpr',g('sgpr_count'),'lds',g('group_segment_fixed_size'),'scratch',g('private_segment_fixed_size'))
PY
