#!/bin/bash
# Build an A/B variant of liboess.so from a patched copy of ONE source file:
#   tools/build_variant.sh <file.hip> <sed-expression> [out.so]      (default out: openess_amd/liboess_b.so)
# Run the two builds on the same box with OESS_LIB_PATH=<out.so> (the only environment variable the wrapper reads).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
F=$1; EXPR=$2; OUT=${3:-$ROOT/openess_amd/liboess_b.so}
make -C $ROOT/openess_amd/csrc -j8 > /dev/null
TMP=$(mktemp -d)
sed -E "$EXPR" $ROOT/openess_amd/csrc/$F > $TMP/$F
if cmp -s $TMP/$F $ROOT/openess_amd/csrc/$F; then echo "sed expression changed nothing"; exit 1; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/openess_amd/csrc -ffp-contract=off \
    -fhip-fp32-correctly-rounded-divide-sqrt -c $TMP/$F -o $TMP/variant.o
OBJS=$(ls $ROOT/openess_amd/csrc/build/*.o | grep -v "/${F%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS $TMP/variant.o
echo "built $OUT"
