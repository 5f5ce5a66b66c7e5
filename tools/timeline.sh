# usage (GPU box): bash tools/timeline.sh <tag> scene spp w h   -- every launch of one render, in order (tools/launch_timeline.py)
TAG=$1; SCENE=$2; SPP=$3; W=$4; H=$5
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 tools/profile_run.py $SCENE $SPP $W $H > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 tools/launch_timeline.py $OUT/trace > $OUT/timeline_${SCENE}_${SPP}.txt
rm -rf $OUT/trace
cat $OUT/timeline_${SCENE}_${SPP}.txt
