#!/bin/bash
# The library with ONE source taken from a git revision (default HEAD), the rest from the working tree's product build -- the
# "before" arm of an A/B in one gpurun call.   usage: tools/dbg/old_lib.sh <source.hip> [rev] [name]  ->  tools/dbg/libs/librn_<name>.so
set -e
cd "$(dirname "$0")/../.."
SRC=$1; REV=${2:-HEAD}; NAME=${3:-old}
P=relationnetworks-clevr_amd
mkdir -p tools/dbg/libs
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
git show $REV:$P/csrc/$SRC > $P/csrc/_old_$SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $P/csrc/_old_$SRC -o tools/dbg/libs/$NAME.o 2>/dev/null || { rm -f $P/csrc/_old_$SRC; exit 1; }
rm -f $P/csrc/_old_$SRC
OBJS=$(ls $P/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libs/librn_$NAME.so $OBJS tools/dbg/libs/$NAME.o
rm tools/dbg/libs/$NAME.o
echo tools/dbg/libs/librn_$NAME.so
