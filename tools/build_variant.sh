#!/bin/bash
# A library variant for A/B runs: one translation unit rebuilt with extra -D flags, linked with the objects of the default
# build into tools/_build/libmsde_<name>.so (git-ignored; travels to the GPU box).  usage: tools/build_variant.sh NAME FILE.hip -DX=1 ...
set -euo pipefail
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
python -m moleculesde_amd.build > /dev/null
mkdir -p tools/_build
B=moleculesde_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value "$@" -c moleculesde_amd/csrc/$SRC -o tools/_build/${SRC%.hip}_$NAME.o
OBJS=$(ls $B/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $OBJS tools/_build/${SRC%.hip}_$NAME.o -o tools/_build/libmsde_$NAME.so
echo tools/_build/libmsde_$NAME.so
