#!/bin/bash
# builds A/B variants of the library: tools/ab_build.sh NAME "EXTRA_FLAGS" -> libxsmm-1_amd/lib/ab/libxsmm_NAME.so
set -e
name=$1; extra=$2
cd "$(dirname "$0")/../libxsmm-1_amd/csrc"
mkdir -p ../lib/ab
make -j8 OUT=../lib/ab/libxsmm_$name.so OBJDIR=../build/ab_$name EXTRA="$extra" 2>&1 | grep -E "error|warning: v|Error" || true
ls -la ../lib/ab/libxsmm_$name.so
