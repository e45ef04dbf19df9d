#!/bin/bash
# PMC comparison of the two main-loop forms of the emulated forward kernel (rocprofv3 --pmc only, one counter group per run).
# usage (repo root, GPU box): tools/pmc_kc2.sh <outdir> [variant libs...]
R=$PWD; O=$R/$1; shift; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TA_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TCC_[A-Z_0-9]+|GRBM_[A-Z_0-9]+)\b" | sort -u > $O/avail.txt
wc -l $O/avail.txt
PMCG=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
        "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU"
        "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
        "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS"
        "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
        "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum"
        "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU")
for cfg in "form1:HOISDF_EMU_KC=1" "$@"; do
  label=${cfg%%:*}; envs=${cfg#*:}
  for case in ${PMC_CASES:-linear_fwd_emu_65536x1024x256 linear_fwd_emu_65536x256x1024}; do
    i=0
    for grp in "${PMCG[@]}"; do
      i=$((i+1))
      rm -rf /tmp/pmc_run
      env $envs timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_run -- python $R/tools/pmc_case.py $case > /tmp/pmc_log.txt 2>&1
      C=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
      if [ -n "$C" ]; then grep -E "Counter_Name|${PMC_KERNEL:-emu_kc}" $C > $O/${label}.${case}.g$i.csv; else echo "no csv for $label $case group $i: $(tail -2 /tmp/pmc_log.txt)"; fi
    done
  done
done
python $R/tools/pmc_kc2_fold.py $O
