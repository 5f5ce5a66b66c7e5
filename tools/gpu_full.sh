# usage (GPU box): bash tools/gpu_full.sh <tag>   -- every GPU test, then the default bench line
TAG=${1:-full}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -u -m pytest tests -x -v -m gpu --timeout 300 --timeout-method thread > gpurun_out/$TAG/pytest.log 2>&1 || { tail -40 gpurun_out/$TAG/pytest.log; exit 1; }
tail -3 gpurun_out/$TAG/pytest.log
timeout -k 10 300 python -u bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { tail -20 gpurun_out/$TAG/bench.err; exit 1; }
cat gpurun_out/$TAG/bench.json
