set -e
MI355RAG_LIB=$PWD/tools/_trace/libtrace.so timeout -k 10 300 python -u tools/trace_scan.py > gpurun_out/trace.log 2>&1
