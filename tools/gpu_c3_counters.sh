#!/bin/bash
# On the GPU box: why is the colour sampler (BASELINE C3, sdfk_sample_bits_clip at 512^3) at 0.71-0.78 of the HBM peak where the sphere's
# sampler reaches 0.83?  L2 -> fabric write counters, address translation and vector-cache stall counters of both, one counter group per
# pass (rocprofv3 --pmc only), + the list of what this rocprofv3 offers.  usage: tools/gpu_c3_counters.sh r05
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/c3_$tag; mkdir -p $O
rocprofv3 -L 2>/dev/null > $O/counters_available.txt
grep -o -E "\b(TCC|TCP|TCA|TA|TD|GRBM|SQ|SQC|GL2C|UTCL2|ATC)[A-Z0-9_]*\b" $O/counters_available.txt | sort -u > $O/counter_names.txt
wc -l $O/counter_names.txt
grep -E "STALL|UTCL|WRREQ|WRITE|EA0" $O/counter_names.txt | tr '\n' ' ' | cut -c1-3000
echo
for scene in repeatxy sphere; do
  python3 tools/c3_sampler_probe.py $scene 512 2 20
  python3 tools/c3_sampler_probe.py $scene 512 4 20
done | tee $O/timing.txt
pass() {   # pass NAME COUNTERS...
  name=$1; shift
  for scene in repeatxy sphere; do
    timeout 300 rocprofv3 --pmc "$@" -d $R/$O/pmc_${name}_$scene -o p --output-format csv -- python3 tools/c3_sampler_probe.py $scene 512 2 6 > /dev/null 2>$O/err_${name}_$scene.txt \
      && python3 tools/pmc_summary.py $O/pmc_${name}_$scene/p_counter_collection.csv 2>/dev/null | grep -A40 "^sdfk_sample_bits" | sed "s/^sdfk_sample_bits[a-z_]*/& [$scene]/" \
      || echo "pass $name ($*) failed for $scene: $(tail -2 $O/err_${name}_$scene.txt | tr '\n' ' ' | cut -c1-300)"
    rm -rf $O/pmc_${name}_$scene
  done
}
{
pass ea_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum
pass ea_credit TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
pass tcc_req TCC_REQ_sum TCC_WRITE_sum TCC_STREAMING_REQ_sum TCC_NC_REQ_sum
pass tcc_busy TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum
pass utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum
pass tcp_stall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum
pass ta TA_BUSY_sum TA_BUFFER_WRITE_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pass sq2 SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SALU
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
} 2>&1 | tee $O/c3_sampler_counters.txt
