set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t6.log 2>&1
timeout -k 10 200 python -u tools/regime_sweep.py > gpurun_out/sweep6.jsonl 2>> gpurun_out/ab.err
