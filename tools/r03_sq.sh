#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
run() { rm -rf gpurun_out/pmc_sq$1; shift_n=$1; shift; timeout -k 10 500 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_sq$shift_n -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq$shift_n.log 2>&1 || { tail -20 gpurun_out/pmc_sq$shift_n.log; exit 1; }; }
run 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run 2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES
run 3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
python tools/pmc_sq.py gpurun_out/r3_pmc_sq.csv $(find gpurun_out/pmc_sq1 -name "*counter_collection.csv") $(find gpurun_out/pmc_sq2 -name "*counter_collection.csv") $(find gpurun_out/pmc_sq3 -name "*counter_collection.csv")
rm -f $(find gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3 -name "*kernel_trace.csv")
