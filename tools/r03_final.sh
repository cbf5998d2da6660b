#!/bin/bash
# round 3 end state: every number quoted in DESIGN.md / profiles/README.md comes from this script's outputs
T=${TAG:-r03_z}
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/$T; mkdir -p $O
step() { echo "== $*"; }
step bench default; timeout -k 10 600 python bench.py --layers > $O/bench_line.json 2> $O/bench_line.err || { tail -20 $O/bench_line.err; exit 1; }
step kernel stats; rm -rf $O/prof; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_line_under_rocprof.json 2> $O/prof.err || { tail -20 $O/prof.err; exit 1; }
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; rm -rf $O/prof
step pmc fetch; rm -rf $O/pf; timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pf.log 2>&1 || { tail -20 $O/pf.log; exit 1; }
step pmc write; rm -rf $O/pw; timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pw.log 2>&1 || { tail -20 $O/pw.log; exit 1; }
python tools/pmc_traffic.py $(find $O/pf -name "*counter_collection.csv") $(find $O/pw -name "*counter_collection.csv") $O/pmc_traffic > /dev/null || exit 1
rm -rf $O/pf $O/pw
step pmc mfma; rm -rf $O/pm; timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pm -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pm.log 2>&1 || { tail -20 $O/pm.log; exit 1; }
python tools/pmc_mfma.py $(find $O/pm -name "*counter_collection.csv") $O/pmc_mfma $O/bench_line.json > /dev/null || exit 1
rm -rf $O/pm
step noise; timeout -k 10 600 python bench.py --no-cpu-baseline --background noise > $O/noise_bench_line.json 2> $O/noise.err || { tail -20 $O/noise.err; exit 1; }
rm -rf $O/prof; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --background noise > $O/noise_bench_line_under_rocprof.json 2> $O/prof.err || { tail -20 $O/prof.err; exit 1; }
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/noise_bench_kernel_stats.csv; rm -rf $O/prof
step first_a; timeout -k 10 600 python bench.py --no-cpu-baseline --approach first_a > $O/first_a_bench_line.json 2> $O/first_a.err || { tail -20 $O/first_a.err; exit 1; }
step first_b; timeout -k 10 600 python bench.py --no-cpu-baseline --approach first_b --steps 500 > $O/first_b_bench_line.json 2> $O/first_b.err || { tail -20 $O/first_b.err; exit 1; }
rm -rf $O/prof; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 100 --warmup 3 --no-cpu-baseline --approach first_b > $O/first_b_bench_line_under_rocprof.json 2> $O/prof.err || { tail -20 $O/prof.err; exit 1; }
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/first_b_bench_kernel_stats.csv; rm -rf $O/prof
step cfg5; timeout -k 10 600 python bench.py --no-cpu-baseline --mesh bob --style 5 --target both --steps 200 > $O/config5_200steps_bench_line.json 2> $O/cfg5.err || { tail -20 $O/cfg5.err; exit 1; }
step cfg3; timeout -k 10 600 python bench.py --no-cpu-baseline --size 1024 --views 16 --style 3 --steps 30 > $O/config3_shape_bench_line.json 2> $O/cfg3.err || { tail -20 $O/cfg3.err; exit 1; }
step cfg4; timeout -k 10 600 python bench.py --no-cpu-baseline --mesh teapot --style 4 --steps 100 > $O/config4_share_bench_line.json 2> $O/cfg4.err || { tail -20 $O/cfg4.err; exit 1; }
step gpus2 gloo; ST3D_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 20 --no-cpu-baseline > $O/gpus2_gloo_one_gpu_line.json 2> $O/gpus2.err || { tail -20 $O/gpus2.err; exit 1; }
python - <<P
import json,glob,os
for f in sorted(glob.glob("$O/*line.json")):
    d=json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"]["frac"], d["step_roofline"]["frac"])
P
