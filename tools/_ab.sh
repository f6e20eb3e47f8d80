set -e
timeout -k 10 600 python -u -m pytest tests/test_search_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t4.log 2>&1
for i in 1 2; do
MI355RAG_LIB=$PWD/tools/_ab/libold.so timeout -k 10 200 python -u tools/regime_sweep.py --cases a > gpurun_out/ab_old_$i.jsonl 2>> gpurun_out/ab.err
timeout -k 10 200 python -u tools/regime_sweep.py --cases a > gpurun_out/ab_new_$i.jsonl 2>> gpurun_out/ab.err
done
