#!/bin/bash
# Registers, LDS and scratch of the kernels of one source file whose (mangled) names match a pattern -- what decides
# their occupancy (512 VGPRs per SIMD lane in granules of 8, 160 KB of LDS per CU).  Runs without a GPU.
# usage: tools/kernel_resources.sh <file under screencounter_amd/csrc> <name regex> [extra hipcc flags]
#   e.g. tools/kernel_resources.sh scg_kernels.hip 'dual_passes_kernelILi5ELi2ELi3'
#        tools/kernel_resources.sh scg_inflate.hip lanes -DSCG_INFLATE_WAVES=7
SRC=$1; PAT=$2; shift; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
TMP=$(mktemp -d)
cd $TMP
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -I$ROOT/include -I$ROOT/screencounter_amd/csrc --save-temps "$@" \
    -c -o k.o $ROOT/screencounter_amd/csrc/$SRC 2>&1 | grep -E "error" -A5
python3 - "$PAT" $TMP <<'PY'
import glob, re, sys
s = open(glob.glob(sys.argv[2] + "/*-hip-amdgcn-amd-amdhsa-gfx950.s")[0]).read()
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size', s, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
    if re.search(sys.argv[1], g('name')):
        print(g('name')[:100], 'vgpr', g('vgpr_count'), 'sgpr', g('sgpr_count'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'))
PY
rm -rf $TMP
