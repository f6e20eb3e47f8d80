set -e
timeout -k 10 500 python -u bench.py > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_a --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_a_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_a_prof.err
