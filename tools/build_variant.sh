#!/bin/bash
# tools/build_variant.sh NAME UNIT "-DFLAG ..." : rebuild one translation unit with extra flags and link it with the library's other
# objects into ab/libblsmi_NAME.so (A/B timing with tools/ab_bench.py; ab/ is scratch, git-ignored)
set -e
cd "$(dirname "$0")/.."
NAME=$1; UNIT=$2; FLAGS=$3
mkdir -p ab
OBJ=ab/${UNIT}_${NAME}.o
EXTRA=""
[ "$UNIT" = "blsmi.hip" ] && EXTRA="-DBLSMI_LAT_BIN=\"$(pwd)/bls_amd/csrc/lat_programs.z\""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Werror=pass-failed $FLAGS $EXTRA -c -o $OBJ bls_amd/csrc/$UNIT
OBJS=""
for o in bls_amd/csrc/build/*.hip.o; do
  if [ "$(basename $o)" = "$UNIT.o" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o ab/libblsmi_$NAME.so $OBJS -lz
echo built ab/libblsmi_$NAME.so
