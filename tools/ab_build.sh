#!/bin/bash
# Builds libscg.so of a git revision into tools/ab/<name>.so (git-ignored) for tools/ab.sh.
# usage: tools/ab_build.sh <git-rev> <name>
set -e
REV=$1; NAME=$2
ROOT=$(cd $(dirname $0)/.. && pwd)
TMP=$(mktemp -d)
git -C $ROOT archive $REV screencounter_amd/csrc include | tar -x -C $TMP
make -s -C $TMP/screencounter_amd/csrc OBJDIR=$TMP/build
mkdir -p $ROOT/tools/ab
cp $TMP/screencounter_amd/libscg.so $ROOT/tools/ab/$NAME.so
rm -rf $TMP
echo "built tools/ab/$NAME.so from $REV"
