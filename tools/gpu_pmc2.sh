#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
export TMPDIR=/tmp SDFK_LANES=0
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM \
    -d $R/gpurun_out/pmcC -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmcC/p_counter_collection.csv
