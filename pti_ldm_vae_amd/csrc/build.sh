#!/bin/bash
# Builds libpti_vae_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [-j N]
set -e
cd "$(dirname "$0")"
OUT=../libpti_vae_hip.so
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffast-math -fno-finite-math-only -Wno-unused-value -Wno-pass-failed"
mkdir -p build
pids=()
for f in *.hip abi.cpp; do
  o=build/${f%.*}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ pti_common.h -nt "$o" ] || [ conv_common.h -nt "$o" ] || [ ../../include/pti_vae.h -nt "$o" ]; then
    ( hipcc $FLAGS -x hip -c "$f" -o "$o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" build/*.o
echo "built $OUT"
