# PMC passes over the bench's scan launches; one counter group per run (rocprofv3 does not split passes).
# usage (on the GPU box): bash tools/pmc_scan.sh [tag] [extra bench args]   -> gpurun_out/pmc_<tag>/p<i>/
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-scan}
shift || true
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_$TAG/p$i --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --embed-chunks 0 "$@" > $R/gpurun_out/pmc_$TAG/p$i.json 2> $R/gpurun_out/pmc_$TAG/p$i.err
  echo "pass $i done"
done
python3 $R/tools/pmc_report.py $R/gpurun_out/pmc_$TAG/p1 $R/gpurun_out/pmc_$TAG/p2 $R/gpurun_out/pmc_$TAG/p3 $R/gpurun_out/pmc_$TAG/p4 --match mfma_scan > $R/gpurun_out/pmc_$TAG/report.txt
cat $R/gpurun_out/pmc_$TAG/p1.json
