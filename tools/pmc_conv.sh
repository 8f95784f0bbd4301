cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM"; do
  d=$R/gpurun_out/pmc3/$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o x -- python $R/tools/conv_microbench.py rn.layer2 conv_arith=1 > /dev/null 2>&1
  python $R/tools/pmc_table.py $(ls $d/*/x_counter_collection.csv $d/x_counter_collection.csv 2>/dev/null | head -1) conv3 conv_wgrad2 conv_split
done
