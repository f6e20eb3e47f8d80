set -e
timeout -k 10 600 python -u -m pytest tests/test_search_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t5.log 2>&1
MI355RAG_LIB=$PWD/tools/_ab/libold.so timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --embed-chunks 0 > gpurun_out/ab100_old_1.json 2>> gpurun_out/ab.err
timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --embed-chunks 0 > gpurun_out/ab100_new_1.json 2>> gpurun_out/ab.err
timeout -k 10 200 python -u tools/regime_sweep.py > gpurun_out/sweep5.jsonl 2>> gpurun_out/ab.err
