#!/bin/bash
# MFMA gather beside the tiled VALU gather on the bench workload (GPU box):  stand-alone durations + SQ / MFMA counters of both
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/serial_stats.sh gm0 --opt pme_weightless=0
bash $R/tools/serial_stats.sh gm1 --opt gather_mfma=1
grep -E "k_pme_gather" $R/gpurun_out/serial_gm0.txt $R/gpurun_out/serial_gm1.txt
PMC_ONLY="k_pme_gather" BENCH_ARGS="--no-thermal --opt overlap=0 --opt graphs=0 --opt gather_mfma=1" bash $R/tools/pmc_gpu.sh gm1 \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE" > /dev/null 2>&1
PMC_ONLY="k_pme_gather" BENCH_ARGS="--no-thermal --opt overlap=0 --opt graphs=0 --opt pme_weightless=0" bash $R/tools/pmc_gpu.sh gm0 \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" > /dev/null 2>&1
cat $R/gpurun_out/pmc_gm1/summary.txt $R/gpurun_out/pmc_gm0/summary.txt
