# usage (GPU box): bash tools/ab_round.sh <tag> "<lib dirs>" "<scene:spp:w:h ...>" [pytest target]
# Parity tests with the in-tree build first, then the same renders timed with each build of the libraries.
TAG=$1; LIBS=$2; CASES=$3; TESTS=${4:-tests/test_gpu_parity.py}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
if [ "$TESTS" != "none" ]; then
  timeout -k 10 600 python -u -m pytest $TESTS -x -q -m gpu --timeout 240 --timeout-method thread > gpurun_out/$TAG/pytest.log 2>&1 || { tail -30 gpurun_out/$TAG/pytest.log; exit 1; }
  tail -2 gpurun_out/$TAG/pytest.log
fi
bash tools/ab_multi.sh $TAG "$LIBS" "$CASES"
