set -e
timeout -k 10 300 python -u -m pytest tests/test_encoder_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/te1.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_enc1 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --embed-only > $GRAFT_REPO_ROOT/gpurun_out/enc1.json 2> $GRAFT_REPO_ROOT/gpurun_out/enc1.err
