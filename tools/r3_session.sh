set -o pipefail
cd $GRAFT_REPO_ROOT
G1="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
G2="SQ_IFETCH SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"
G4="SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
bash tools/pmc_quick.sh r3_pmc_shade_sorted "--solo --spp 64" "$G1" "$G2" "$G3" "$G4"
PTR_SHADE_SORT=0 bash tools/pmc_quick.sh r3_pmc_shade_unsorted "--solo --spp 64" "$G1" "$G2" "$G3" "$G4"
grep -A40 "k_shade_sorted<false" gpurun_out/r3_pmc_shade_sorted.txt | head -45
grep -A40 "k_shade<false, false, false, false" gpurun_out/r3_pmc_shade_unsorted.txt | head -45
