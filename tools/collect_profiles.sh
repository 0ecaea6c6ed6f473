#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun) from the repo root:
#   bash tools/collect_profiles.sh r02 [cfg4|cfg4bf16|cfg3|cfg5]
# rocprofv3 passes over `python3 bench.py` (kernel trace + stats, then PMC passes each in their own
# run, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE cannot share
# a pass).  Raw CSVs land under gpurun_out/prof_<tag>/; tools/summarize_profiles.py turns them into
# the files committed under profiles/.
set -euo pipefail
tag="${1:-r01}"
cfg="${2:-cfg4}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/prof_$tag"
rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
CMD=(python3 "$R/bench.py" --config "$cfg" --steps 20 --warmup 5 --no-cpu-baseline --sustained 0)
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- "${CMD[@]}" > "$O/kt.log" 2>&1
# per-dispatch durations of an un-profiled-counter run with the sustained tail (the post-idle transient): kernel trace only
rocprofv3 --kernel-trace --output-format csv -d "$O/kt_long" -- python3 "$R/bench.py" --config "$cfg" --steps 20 --warmup 5 --no-cpu-baseline --sustained 150 > "$O/kt_long.log" 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --output-format csv -d "$O/pmc_sq" -- "${CMD[@]}" > "$O/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$O/pmc_mfma" -- "${CMD[@]}" > "$O/pmc_mfma.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- "${CMD[@]}" > "$O/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- "${CMD[@]}" > "$O/pmc_write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$O/pmc_l2" -- "${CMD[@]}" > "$O/pmc_l2.log" 2>&1 || true
grep -h '"metric"' "$O"/kt.log | tail -1 > "$O/bench_line_under_kernel_trace.json" || true
echo "profiles collected under $O"
