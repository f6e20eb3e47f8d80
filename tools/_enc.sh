set -e
timeout -k 10 300 python -u -m pytest tests/test_encoder_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/te3.log 2>&1
for i in 1 2; do
MI_GEMM_LN=0 timeout -k 10 200 python -u bench.py --embed-only --no-cpu-baseline > gpurun_out/enc_noln_$i.json 2>> gpurun_out/enc.err
timeout -k 10 200 python -u bench.py --embed-only --no-cpu-baseline > gpurun_out/enc_ln_$i.json 2>> gpurun_out/enc.err
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_enc4 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --embed-only > $GRAFT_REPO_ROOT/gpurun_out/enc4.json 2> $GRAFT_REPO_ROOT/gpurun_out/enc4.err
