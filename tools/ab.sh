# usage: bash tools/ab.sh <tag> [pytest-target]   (on the GPU box: parity tests, then timings of the three named scenes)
set -e
TAG=$1; TESTS=${2:-tests/test_gpu_parity.py}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 400 python -u -m pytest $TESTS -x -q -m gpu --timeout 240 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 120 python -u tools/perf_sweep.py book_cover 64 1920 1080 ${AB_SWEEP:-X=0} > $OUT/perf_cover.txt 2>&1
timeout -k 10 120 python -u tools/perf_sweep.py cornell 64 1024 1024 > $OUT/perf_cornell.txt 2>&1
timeout -k 10 120 python -u tools/perf_sweep.py book2 64 1920 1080 > $OUT/perf_book2.txt 2>&1
cat $OUT/perf_cover.txt $OUT/perf_cornell.txt $OUT/perf_book2.txt
