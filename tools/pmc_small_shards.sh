set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r05_pmc_small
mkdir -p $R
export AMP_BENCH_NO_CALIBRATION=1
for N in 4096 8192; do
ARGS="--envs $N --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --no-fp32-engine --no-dropin --no-configs --no-update"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/sq_$N -- python3 bench.py $ARGS > $R/sq_$N.json 2> $R/sq_$N.err || exit 2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/fetch_$N -- python3 bench.py $ARGS > $R/fetch_$N.json 2> $R/fetch_$N.err || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/write_$N -- python3 bench.py $ARGS > $R/write_$N.json 2> $R/write_$N.err || exit 4
python3 tools/pmc_summary.py $R/sq_$N/*/*_counter_collection.csv $R/fetch_$N/*/*_counter_collection.csv $R/write_$N/*/*_counter_collection.csv > $R/pmc_$N.md
done
