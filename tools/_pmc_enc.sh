set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --embed-only --no-cpu-baseline --embed-steps 1 --embed-chunks 1024"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmce2_sq -- $B > $R/gpurun_out/pmce2_sq.json 2> $R/gpurun_out/pmce2_sq.err
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmce2_lds -- $B > $R/gpurun_out/pmce2_lds.json 2> $R/gpurun_out/pmce2_lds.err
