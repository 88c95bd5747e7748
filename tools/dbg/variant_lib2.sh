#!/bin/bash
# A variant build with extra -D flags on SEVERAL sources: tools/dbg/variant_lib2.sh <name> "<src1.hip:flags>" "<src2.hip:flags>" ...
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
P=relationnetworks-clevr_amd
mkdir -p tools/dbg/libs
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
OBJS=$(ls $P/build/*.o)
EXTRA=""
for spec in "$@"; do
  SRC=${spec%%:*}; FLAGS=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c $P/csrc/$SRC -o tools/dbg/libs/${NAME}_${SRC%.hip}.o 2>/dev/null
  OBJS=$(echo "$OBJS" | grep -v "/${SRC%.hip}.o")
  EXTRA="$EXTRA tools/dbg/libs/${NAME}_${SRC%.hip}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libs/librn_$NAME.so $OBJS $EXTRA
rm -f $EXTRA
echo tools/dbg/libs/librn_$NAME.so
