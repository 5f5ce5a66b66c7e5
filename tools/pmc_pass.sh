# usage: bash tools/pmc_pass.sh <tag> "<counters>" [scene] [spp]   (one rocprofv3 counter pass, on the GPU box)
set -e
TAG=$1; PASS=$2; SCENE=${3:-book_cover}; SPP=${4:-16}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc $PASS -d $OUT/pmc_x -o pmc -- python3 tools/profile_run.py $SCENE $SPP > $OUT/run.log 2>&1
python3 tools/prof_summary.py $OUT > $OUT/summary.txt
