#!/bin/bash
# round 3, first GPU trip: new launcher tests, the 1-GPU line under the corrected accounting, the MFMA counter pass
set -x
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r3_dist.log 2>&1 || { tail -30 gpurun_out/r3_dist.log; exit 1; }
tail -3 gpurun_out/r3_dist.log
timeout -k 10 600 python bench.py --layers > gpurun_out/r3a_line.json 2> gpurun_out/r3a_line.err || { tail -20 gpurun_out/r3a_line.err; exit 1; }
python - <<'P'
import json; d=json.load(open("gpurun_out/r3a_line.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["first_step_loss"], d.get("cpu_baseline"))
P
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_mfma
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_mfma.log 2>&1 || { tail -20 gpurun_out/pmc_mfma.log; exit 1; }
f=$(find gpurun_out/pmc_mfma -name "*counter_collection.csv" | head -1)
python tools/pmc_mfma.py "$f" gpurun_out/r3a_pmc_mfma gpurun_out/r3a_line.json | tail -60
