#!/bin/bash
# build_mw.sh M N K F64 WPE [extra -D...]: compiles tools/probe/mfma_wave.hip for one configuration, prints its register use
cd "$(dirname "$0")" || exit 1
mkdir -p bin tmp
name=mw_$1x$2x$3_$4_$5$(echo "${@:6}" | tr -d ' -' )
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -DXM=$1 -DXN=$2 -DXK=$3 -DXF64=$4 -DXWPE=$5 "${@:6}" -save-temps=obj mfma_wave.hip -o tmp/$name 2>&1 | grep -i "error"
mv tmp/$name bin/$name
echo "$name: $(grep -E '^\s+\.(vgpr_count|agpr_count|vgpr_spill_count)' tmp/mfma_wave-hip-amdgcn-amd-amdhsa-gfx950.s | tr -s ' ' | tr '\n' ' ')"
