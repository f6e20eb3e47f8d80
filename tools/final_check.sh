# End-of-round check on the GPU box:  gpurun -- 'bash tools/final_check.sh'
# smoke -> all GPU tests -> the bench line -> the same bench under rocprofv3 --kernel-trace --stats (gpurun_out/prof_final).
set -e
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_all.log 2>&1
timeout -k 10 500 python -u bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_final_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_final_prof.err
