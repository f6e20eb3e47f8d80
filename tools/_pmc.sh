set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --embed-chunks 0"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc2_fetch -- $B > $R/gpurun_out/pmc2_fetch.json 2> $R/gpurun_out/pmc2_fetch.err
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc2_write -- $B > $R/gpurun_out/pmc2_write.json 2> $R/gpurun_out/pmc2_write.err
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc2_sq -- $B > $R/gpurun_out/pmc2_sq.json 2> $R/gpurun_out/pmc2_sq.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench_b_prof.json 2> $R/gpurun_out/bench_b_prof.err
