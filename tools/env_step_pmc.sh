#!/bin/bash
# Run ON THE GPU BOX: instruction-mix PMC passes of the env-step launch alone (tools/env_step_bench.py).
#   gpurun -- 'bash tools/env_step_pmc.sh r02_env 65536 g1_walk'
set -o pipefail
TAG=${1:-env_pmc}; ENVS=${2:-65536}; WL=${3:-g1_walk}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/$TAG
mkdir -p "$R"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM \
  --output-format csv -d "$R/p1" -- python3 tools/env_step_bench.py $ENVS $WL > "$R/p1.json" 2> "$R/p1.err" || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d "$R/p2" -- python3 tools/env_step_bench.py $ENVS $WL > "$R/p2.json" 2> "$R/p2.err" || exit 2
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d "$R/p3" -- python3 tools/env_step_bench.py $ENVS $WL > "$R/p3.json" 2> "$R/p3.err" || exit 3
python3 tools/pmc_summary.py "$R"/p1/*/*_counter_collection.csv "$R"/p2/*/*_counter_collection.csv "$R"/p3/*/*_counter_collection.csv > "$R/pmc_per_kernel.md"
cat "$R/pmc_per_kernel.md"
