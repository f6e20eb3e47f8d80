set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_enc --output-format csv -- python3 $R/bench.py --embed-only --no-cpu-baseline --embed-steps 1 > $R/gpurun_out/pmc_enc.json 2> $R/gpurun_out/pmc_enc.err
