set -e
timeout -k 10 600 python -u -m pytest tests/test_search_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t11.log 2>&1
MI_MFMA_M16=0 timeout -k 10 200 python -u tools/regime_sweep.py > gpurun_out/sweep_m32.jsonl 2>> gpurun_out/ab.err
timeout -k 10 200 python -u tools/regime_sweep.py > gpurun_out/sweep_m16.jsonl 2>> gpurun_out/ab.err
