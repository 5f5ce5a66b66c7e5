# usage (GPU box): bash tools/bench_trace.sh <tag>   -- rocprofv3 kernel trace + stats of the bench command itself
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_bench
mkdir -p $OUT gpurun_out/$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/$TAG/bench_under_rocprof.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu   (1 warm-up + 3 timed frames + the instrumented pass)"; echo "# bench line of the same run:"; echo "# $(cat gpurun_out/$TAG/bench_under_rocprof.json | cut -c1-400)"; python3 tools/prof_summary.py $OUT; } > gpurun_out/$TAG/${TAG}_bench_kernel_trace.txt
rm -rf $OUT
head -20 gpurun_out/$TAG/${TAG}_bench_kernel_trace.txt
