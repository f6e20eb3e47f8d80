#!/bin/bash
# Builds the cycle-stamp debug library of the scan (tools/_trace/libtrace.so): search_mfma.hip with -DMF_TRACE
# and encoder.hip with -DFF_TRACE (extra flags in $1, e.g. -DMF_TRACE_U0=300), the other objects as built by rag_amd/_build.py.
set -e
cd "$(dirname "$0")/.."
python -m rag_amd._build > /dev/null
mkdir -p tools/_trace
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Wno-inline-asm"
/opt/rocm/bin/hipcc $F -DMF_TRACE $1 -c rag_amd/csrc/search_mfma.hip -o tools/_trace/search_mfma.o
/opt/rocm/bin/hipcc $F -DFF_TRACE $1 -c rag_amd/csrc/encoder.hip -o tools/_trace/encoder.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_trace/libtrace.so build/obj/capi.o build/obj/multi.o build/obj/ingest.o build/obj/search_exact.o build/obj/encoder_ffn3.o build/obj/encoder_qkv_attn.o build/obj/wordpiece.o tools/_trace/encoder.o tools/_trace/search_mfma.o
echo built tools/_trace/libtrace.so
