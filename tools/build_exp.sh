#!/bin/bash
# build an experiment variant of the library into .exp/<name>/ with extra -D flags for ONE translation unit.  usage: build_exp.sh <name> <unit.hip> <flags...>
cd /root/repo
n=$1; unit=$2; shift 2
mkdir -p .exp/$n
cp nvfi_amd/csrc/*.o .exp/$n/ && touch .exp/$n/*.o && rm -f .exp/$n/${unit%.hip}.o
NVFI_BUILD_SO=$PWD/.exp/$n/libnvfi_hip.so NVFI_BUILD_OBJDIR=$PWD/.exp/$n NVFI_EXTRA_FLAGS="$*" python -m nvfi_amd.build | tail -1
