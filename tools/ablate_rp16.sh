#!/usr/bin/env bash
# On the GPU box: cycle counters of the shipped kernel (algo 24) for each timing-ablation build in gpurun_variants/
# (tools/build_variant.sh ablN "-DFA_RP16_ABL=N -DFA_RP16_GATES=0").   bash tools/ablate_rp16.sh "base abl1 abl2 ..."
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$R/gpurun_out/ablation_rp16.txt"
echo "# fa_fwd_rp16 folded (algo 24) timing ablations, B8 H16 N4096 d64 fp16: rocprofv3 cycle counters per build (FA_RP16_ABL bits: 1 no LDS fragment reads, 2 no softmax vector work, 4 no matrix instructions, 8 no K/V staging, 16 no tile barrier; results of ablated builds are garbage by construction)" > "$out"
for v in $1; do
    echo "== $v" >> "$out"
    FA_MI355_LIB="$R/gpurun_variants/lib_$v.so" bash "$R/tools/pmc_algos.sh" "abl_$v" 24 f16 | grep rp16 >> "$out"
done
cat "$out"
