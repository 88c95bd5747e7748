#!/bin/bash
# A variant build of the library: one source recompiled with extra -D flags, linked with the product build's other objects.
# usage: tools/dbg/variant_lib.sh <name> <source.hip> <flags...>   ->  tools/dbg/libs/librn_<name>.so
set -e
cd "$(dirname "$0")/../.."
NAME=$1; SRC=$2; shift 2
P=relationnetworks-clevr_amd
mkdir -p tools/dbg/libs
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $P/csrc/$SRC -o tools/dbg/libs/$NAME.o 2>/dev/null
OBJS=$(ls $P/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libs/librn_$NAME.so $OBJS tools/dbg/libs/$NAME.o
rm tools/dbg/libs/$NAME.o
echo tools/dbg/libs/librn_$NAME.so
