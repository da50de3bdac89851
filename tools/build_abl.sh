#!/bin/bash
# tools/build_abl.sh <macro> <values...>: liboess_<macro>_<v>.so with -D<macro>=<v> on conv_fwd.hip (ablation builds of conv_lstm_w128.h)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
M=$1; shift
for v in "$@"; do
  ( T=$(mktemp -d); /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/openess_amd/csrc -ffp-contract=off \
      -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-lambda-capture -D$M=$v $EXTRA -c $ROOT/openess_amd/csrc/conv_fwd.hip -o $T/v.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/openess_amd/liboess_${M}_$v$SUFFIX.so $(ls $ROOT/openess_amd/csrc/build/*.o | grep -v /conv_fwd.o) $T/v.o && echo built $v ) &
done
wait
