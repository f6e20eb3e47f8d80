# PMC passes over one bulk-embedding call (4096 chunks x 256 tokens, MiniLM-L6), one counter group per run.
# usage (on the GPU box): bash tools/pmc_encoder.sh   -> gpurun_out/pmc_enc/p<i>/
set -e
export MI_ENC_LANES=1   # one lane: kernels of different passes must not overlap while counters are attributed per dispatch
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_enc
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_enc/p$i --output-format csv -- python3 $R/tools/embed_once.py > $R/gpurun_out/pmc_enc/p$i.json 2> $R/gpurun_out/pmc_enc/p$i.err
  echo "pass $i done"
done
python3 $R/tools/pmc_report.py $R/gpurun_out/pmc_enc/p1 $R/gpurun_out/pmc_enc/p2 $R/gpurun_out/pmc_enc/p3 > $R/gpurun_out/pmc_enc/report.txt
