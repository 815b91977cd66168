#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace + the three PMC passes of the default bench.py command.
#   gpurun -- 'bash tools/collect_profiles.sh r01_b'
# Counters are collected in their own runs (no --stats / trace domains next to --pmc).
set -o pipefail
# (round 5: bench.py prints per-kernel detail on stderr; the summaries below read the rocprofv3 CSVs only)
TAG=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/$TAG
mkdir -p "$R"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/trace" -- python3 bench.py > "$R/bench_traced.json" 2> "$R/bench_traced.err" || exit 1
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-fp32-engine --no-dropin --no-configs"
export AMP_BENCH_NO_CALIBRATION=1  # the PMC passes need only the step kernels
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --output-format csv -d "$R/pmc_sq" -- python3 bench.py $ARGS > "$R/pmc_sq.json" 2> "$R/pmc_sq.err" || exit 2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/pmc_fetch" -- python3 bench.py $ARGS > "$R/pmc_fetch.json" 2> "$R/pmc_fetch.err" || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/pmc_write" -- python3 bench.py $ARGS > "$R/pmc_write.json" 2> "$R/pmc_write.err" || exit 4
unset AMP_BENCH_NO_CALIBRATION
python3 bench.py > "$R/bench_plain.json" 2> "$R/bench_plain.err" || exit 5
python3 tools/prof_summary.py "$R"/trace/*/*_kernel_trace.csv > "$R/kernel_trace_by_grid.md"
python3 tools/pmc_summary.py "$R"/pmc_sq/*/*_counter_collection.csv "$R"/pmc_fetch/*/*_counter_collection.csv "$R"/pmc_write/*/*_counter_collection.csv > "$R/pmc_per_kernel.md"
python3 tools/pmc_summary.py --traffic-json "$R/pmc_traffic.json" "$R"/pmc_fetch/*/*_counter_collection.csv "$R"/pmc_write/*/*_counter_collection.csv
cp "$R"/trace/*/*_kernel_stats.csv "$R/kernel_stats.csv" 2>/dev/null
echo "profiles collected under $R"
