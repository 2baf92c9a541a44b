#!/bin/bash
# build an A/B variant of the library: tools/build_variant.sh NAME "-DFLAG=.. -DFLAG2" -> veritasfi_amd/lib/libvf_NAME.so (select with VF_LIB_PATH)
set -e
name=$1; shift
VF_BUILD_LIB=libvf_$name.so VF_BUILD_TAG=_$name VF_BUILD_FLAGS="$*" python -m veritasfi_amd.build 2>&1 | grep -E "error|libvf_" | tail -3
