#!/bin/bash
# A/B library: search_mfma.hip (or the file named in $FILE) built with the extra flags in $2 into tools/_ab/$1.so;
# run it through MI355RAG_LIB=tools/_ab/$1.so
set -e
cd "$(dirname "$0")/.."
python -m rag_amd._build > /dev/null
mkdir -p tools/_ab
FILE=${FILE:-search_mfma}
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Wno-inline-asm"
if [ "$FILE" == "encoder_ffn3" ]; then F="$F -mllvm -amdgpu-spill-vgpr-to-agpr=0"; fi   # as rag_amd/_build.py FILE_FLAGS
/opt/rocm/bin/hipcc $F $2 -c rag_amd/csrc/$FILE.hip -o tools/_ab/$1_$FILE.o
OBJS=""
for o in capi multi ingest search_exact search_mfma encoder encoder_ffn3 encoder_qkv_attn wordpiece; do
  if [ "$o" == "$FILE" ]; then OBJS="$OBJS tools/_ab/$1_$FILE.o"; else OBJS="$OBJS build/obj/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/$1.so $OBJS
echo built tools/_ab/$1.so
