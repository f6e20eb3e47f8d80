set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --embed-only --no-cpu-baseline --embed-steps 1 --embed-chunks 1024"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmce_sq -- $B > $R/gpurun_out/pmce_sq.json 2> $R/gpurun_out/pmce_sq.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmce_fetch -- $B > $R/gpurun_out/pmce_fetch.json 2> $R/gpurun_out/pmce_fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmce_write -- $B > $R/gpurun_out/pmce_write.json 2> $R/gpurun_out/pmce_write.err
