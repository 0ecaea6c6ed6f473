#!/bin/bash
# Build a variant of the product library with extra -D flags into gpurun_variants/lib_<name>.so (A/B with tools/ab_libs.py).
#   tools/build_variant.sh prio1 "-DFA_SK_PRIO=1"
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/flashattention_kernel_project_amd/csrc
out=$root/gpurun_variants; obj=$out/obj_$name
mkdir -p "$obj"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -fvisibility=hidden -Wall -Wno-unused-function"
SRCS="fa_fwd_kernels.hip fa_fwd_il.hip fa_fwd_rp16.hip fa_fwd_rp16_d64.hip fa_fwd_rp16_d64n.hip fa_fwd_rp16_d64ks.hip fa_fwd_rp16_d128.hip fa_fwd_rp16_d128w.hip fa_fwd_rp16_c.hip fa_fwd_rp16_cw.hip fa_fwd_split.hip fa_debug_stages.hip fa_streaming16.hip fa_capi.hip"
pids=()
for s in $SRCS; do
  # only fa_fwd_sk.hip and fa_capi.hip depend on the knobs in practice; the rest are reused from the product build
  # ONLY=<file.hip> rebuilds exactly that translation unit; otherwise only the d = 64 full-width family depends on the FA_RP16_* knobs in practice (ALL=1 rebuilds every rp16 family); the rest are reused from the product build
  if [ -n "$ONLY" ]; then want=$([ "$s" = "$ONLY" ] && echo 1 || echo 0); else want=-1; fi
  if [ "$want" = 1 ] || [ ! -f "$src/${s%.hip}.o" ]; then
    /opt/rocm/bin/hipcc $FLAGS $extra -c "$src/$s" -o "$obj/${s%.hip}.o" &
    pids+=($!)
  elif [ "$want" = 0 ]; then
    cp "$src/${s%.hip}.o" "$obj/${s%.hip}.o"
  elif [ "$s" = "fa_fwd_rp16_d64.hip" ] || { [ -n "$ALL" ] && [[ "$s" == fa_fwd_rp16_* ]]; } || [ ! -f "$src/${s%.hip}.o" ]; then
    /opt/rocm/bin/hipcc $FLAGS $extra -c "$src/$s" -o "$obj/${s%.hip}.o" &
    pids+=($!)
  else
    cp "$src/${s%.hip}.o" "$obj/${s%.hip}.o"
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $obj/*.o -o "$out/lib_$name.so"
echo "built $out/lib_$name.so"
