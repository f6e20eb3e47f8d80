set -e
timeout -k 10 600 python -u -m pytest tests/test_search_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t9.log 2>&1
for i in 1 2; do
MI355RAG_LIB=$PWD/tools/_ab/libprev.so timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --embed-chunks 0 > gpurun_out/ab100_prev_$i.json 2>> gpurun_out/ab.err
timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --embed-chunks 0 > gpurun_out/ab100_new_$i.json 2>> gpurun_out/ab.err
done
