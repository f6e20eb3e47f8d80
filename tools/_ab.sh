set -e
timeout -k 10 400 python -u -m pytest tests/test_search_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_search.log 2>&1
MI355RAG_LIB=$PWD/tools/_ab/libprev.so timeout -k 10 300 python -u tools/regime_sweep.py --cases a > gpurun_out/ab_prev.jsonl 2> gpurun_out/ab_prev.err
timeout -k 10 300 python -u tools/regime_sweep.py --cases a > gpurun_out/ab_new.jsonl 2> gpurun_out/ab_new.err
MI355RAG_LIB=$PWD/tools/_ab/libprev.so timeout -k 10 300 python -u tools/regime_sweep.py --cases c,d > gpurun_out/ab_prev_cd.jsonl 2> gpurun_out/ab_prev.err
timeout -k 10 300 python -u tools/regime_sweep.py --cases c,d > gpurun_out/ab_new_cd.jsonl 2> gpurun_out/ab_new.err
