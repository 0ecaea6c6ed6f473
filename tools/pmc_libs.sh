#!/usr/bin/env bash
# On the GPU box: cycle + LDS counters of one algo id for several builds in gpurun_variants/ (two rocprofv3 --pmc passes per build).
#   bash tools/pmc_libs.sh <tag> "<name> <name> ..." [algo] [dtype] [extra ab_bench args]
set -euo pipefail
tag="$1"; names="$2"; algo="${3:-24}"; dtype="${4:-f16}"; extra="${5:-}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$R/gpurun_out/pmc_libs_$tag.txt"
: > "$out"
cd /tmp; export TMPDIR=/tmp
for v in $names; do
    O="$R/gpurun_out/pmc_libs_${tag}_$v"; rm -rf "$O"; mkdir -p "$O"
    export FA_MI355_LIB="$R/gpurun_variants/lib_$v.so"
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES \
        --output-format csv -d "$O/a" -- python3 "$R/tools/ab_bench.py" --algos "$algo" --rounds 3 --iters 10 --dtype "$dtype" $extra > "$O/a.log" 2>&1
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_SALU \
        --output-format csv -d "$O/b" -- python3 "$R/tools/ab_bench.py" --algos "$algo" --rounds 3 --iters 10 --dtype "$dtype" $extra > "$O/b.log" 2>&1
    echo "== $v" >> "$out"
    python3 "$R/tools/pmc_summary.py" "$O/a" >> "$out"
    python3 "$R/tools/pmc_summary.py" "$O/b" --lds >> "$out"
    rm -rf "$O"
done
cat "$out"
