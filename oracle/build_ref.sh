#!/usr/bin/env bash
# oracle/build_ref.sh -- build the reference's OWN general-shape CPU oracle from where it lies.
#
# TEST INFRASTRUCTURE ONLY.  Runs only in the build container (where /root/reference exists);
# the GPU box uses the prebuilt oracle/_ref/libref_cpu.so that travels with the snapshot.
#
# The reference is 71 stand-alone .cu programs; every one includes <cuda_runtime.h> and defines
# __global__ kernels, so no file compiles as a whole without nvcc (absent here) -- the device
# code and the per-file main() are unbuildable in this image and we do not try.  What IS plain
# host C++ is the CPU oracle function itself:
#     GEMM/FlashAttention Forward Fused/flashattn_forward_fused_5_4_2.cu  `flashattn_cpu_ref`
# It uses only <vector>, <cmath>, <algorithm> (all of which that file itself includes).  This
# script streams that one function's line range straight from /root/reference into g++ (stdin),
# followed by a C-ABI shim; no reference text is written into the repository and nothing is
# stubbed: no stand-in headers, libraries or generated code are involved.
#
# The 16x16 family's oracle (Streaming_FlashAttention_Forward_Kernel/flashattn_streaming_16x16_mw.cu
# :252-317) needs <cuda_fp16.h>'s __half, which this image lacks, so it is NOT built; our
# restatement of it is cross-checked against this general oracle at D=16 instead (tests/).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ref_root="${FA_REFERENCE_ROOT:-/root/reference}"
src="$ref_root/GEMM/FlashAttention Forward Fused/flashattn_forward_fused_5_4_2.cu"
out_dir="$here/_ref"
if [[ ! -f "$src" ]]; then
    echo "build_ref: reference not present at $ref_root; keeping any prebuilt $out_dir" >&2
    exit 0
fi
mkdir -p "$out_dir"
{
    printf '#include <vector>\n#include <cmath>\n#include <algorithm>\n#include <cstddef>\n'
    # from the function's signature line to the first closing brace in column 0
    awk '/^void flashattn_cpu_ref\(/{on=1} on{print} on&&/^}/{exit}' "$src"
    cat "$here/ref_shim.inc"
} | g++ -O2 -std=c++17 -fPIC -shared -x c++ - -o "$out_dir/libref_cpu.so"
echo "build_ref: built $out_dir/libref_cpu.so from $src"
