#!/bin/bash
# tools/build_variant.sh NAME UNIT[,UNIT...] "-DFLAG ..." : rebuild the named translation units with extra flags and link them with the library's other
# objects into ab/libblsmi_NAME.so (A/B timing through BLSMI_LIB, e.g. tools/ab_bench.py; ab/ is scratch, git-ignored)
set -e
cd "$(dirname "$0")/.."
NAME=$1; UNITS=$2; FLAGS=$3
mkdir -p ab
OBJS=""
for o in bls_amd/csrc/build/*.hip.o; do
  U=$(basename $o .o)
  if [[ ",$UNITS," == *",$U,"* ]]; then
    OBJ=ab/${U}_${NAME}.o
    EXTRA=$(cat $o.flags | tr ' ' '\n' | grep -- '-DBLSMI_LIMBS28' || true)
    [ "$U" = "blsmi.hip" ] && EXTRA="$EXTRA -DBLSMI_LAT_BIN=\"$(pwd)/bls_amd/csrc/lat_programs.z\""
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Werror=pass-failed $FLAGS $EXTRA -c -o $OBJ bls_amd/csrc/$U &
    OBJS="$OBJS $OBJ"
  else
    OBJS="$OBJS $o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o ab/libblsmi_$NAME.so $OBJS -lz
echo built ab/libblsmi_$NAME.so
