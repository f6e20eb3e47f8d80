#!/bin/bash
# Round profile: rocprofv3 --kernel-trace --stats of (1) the default bench command, (2) the embed-only run and
# (3) the 768-d / k = 100 scan; summaries land in gpurun_out/<tag>_*/ for copying into profiles/.
# Usage (on the GPU box): bash tools/profile_round.sh r02
set -e -o pipefail
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
run() {   # name, bench args...
    name=$1; shift
    rm -rf $out/${tag}_$name
    (cd /tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_$name -o p -- python3 $root/bench.py "$@" > $out/${tag}_$name.json 2> $out/${tag}_$name.err)
    tail -n 1 $out/${tag}_$name.json
    find $out/${tag}_$name -name '*kernel_stats.csv' -exec cp {} $out/${tag}_${name}_kernel_stats.csv \;
}
run bench
MI_ENC_LANES=1 run embed --embed-only --no-cpu-baseline      # one lane: per-kernel durations (two lanes overlap kernels of different passes)
run embed_lanes --embed-only --no-cpu-baseline
run scan768 --rows 50000000 --dim 768 --k 100 --no-cpu-baseline --embed-chunks 0
timeout -k 10 300 python3 $root/bench.py > $out/${tag}_bench_plain.json 2> $out/${tag}_bench_plain.err
tail -n 1 $out/${tag}_bench_plain.json
