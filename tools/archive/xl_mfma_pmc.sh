# counters of the matrix-core extract_links kernels (one counter group per pass; FETCH_SIZE and WRITE_SIZE in passes of their own): bash tools/xl_mfma_pmc.sh L TR
R=$GRAFT_REPO_ROOT
for c in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  echo "## $c"
  timeout 300 bash $R/tools/pmc_any.sh "$c" xl_mfma $R/tools/xl_mfma_prof.py $1 $2
done
