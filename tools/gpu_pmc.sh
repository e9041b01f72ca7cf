#!/bin/bash
# On the GPU box: two PMC passes over a short serial (SDFK_LANES=0) bench; per-kernel means.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
export TMPDIR=/tmp SDFK_LANES=0
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS \
    -d $R/gpurun_out/pmcA -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --minimal > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU \
    -d $R/gpurun_out/pmcB -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --minimal > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmcA/p_counter_collection.csv gpurun_out/pmcB/p_counter_collection.csv
