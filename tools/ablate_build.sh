#!/bin/bash
# Builds the measurement-only library tools/ablate/libscg_ablate.so (git-ignored; travels with gpurun): the working
# tree's libscg compiled with -DSCG_ABLATE, the only build in which the SCG_ABLATE environment variable
# (phase ablation: wrong counts by design) has any effect.  Select it with SCG_LIB=tools/ablate/libscg_ablate.so.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/tools/ablate
make -s -C $ROOT/screencounter_amd/csrc OUT=$ROOT/tools/ablate/libscg_ablate.so EXTRA=-DSCG_ABLATE OBJDIR=$ROOT/tools/ablate/build
echo "built tools/ablate/libscg_ablate.so"
