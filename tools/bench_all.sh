# usage (GPU box): bash tools/bench_all.sh <tag> [config ...]   -- bench.py --config for each named configuration
TAG=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
for c in ${*:-book_cover cornell book2 cover1m}; do
  t0=$(date +%s)
  timeout -k 10 580 python -u bench.py --config $c > gpurun_out/$TAG/bench_$c.json 2> gpurun_out/$TAG/bench_$c.err || { echo "$c failed"; tail -5 gpurun_out/$TAG/bench_$c.err; exit 1; }
  echo "$c: $(( $(date +%s) - t0 )) s"; cut -c1-400 gpurun_out/$TAG/bench_$c.json
done
