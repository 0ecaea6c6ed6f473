#!/usr/bin/env bash
# On the GPU box: per-kernel cycle counters for a set of explicit algo ids (one rocprofv3 --pmc pass).
#   bash tools/pmc_algos.sh <tag> <algos> [dtype]      e.g.  bash tools/pmc_algos.sh a1 16,14,17 f16
set -euo pipefail
tag="$1"; algos="$2"; dtype="${3:-f16}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/pmc_$tag"
rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES \
    --output-format csv -d "$O/pmc" -- python3 "$R/tools/ab_bench.py" --algos "$algos" --rounds 3 --iters 10 --dtype "$dtype" > "$O/run.log" 2>&1
python3 "$R/tools/pmc_summary.py" "$O/pmc" | tee "$O/summary.txt"
