#!/bin/bash
# rocprofv3 kernel stats of `bench.py --embed-only` with one encoder lane (a kernel's duration is then its own).
# Usage (on the GPU box): bash tools/prof_embed.sh <tag> [ENV=VALUE ...]
set -e -o pipefail
tag=${1:-embed}; shift || true
for kv in "$@"; do export "$kv"; done
export MI_ENC_LANES=1 TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out
mkdir -p $out; rm -rf $out/prof_$tag
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o p -- python3 $root/bench.py --embed-only --no-cpu-baseline > $out/prof_$tag.json 2> $out/prof_$tag.err)
find $out/prof_$tag -name '*kernel_stats.csv' -exec cp {} $out/prof_${tag}_kernel_stats.csv \;
find $out/prof_$tag -name '*kernel_trace.csv' -exec rm {} \;
head -12 $out/prof_${tag}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
