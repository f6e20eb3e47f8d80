set -e
timeout -k 10 300 python -u -m pytest tests/test_encoder_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/te4.log 2>&1
for i in 1 2; do
MI355RAG_LIB=$PWD/tools/_ab/libprev.so timeout -k 10 200 python -u bench.py --embed-only --no-cpu-baseline > gpurun_out/enc_prev_$i.json 2>> gpurun_out/enc.err
timeout -k 10 200 python -u bench.py --embed-only --no-cpu-baseline > gpurun_out/enc_new_$i.json 2>> gpurun_out/enc.err
done
