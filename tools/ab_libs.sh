# usage: bash tools/ab_libs.sh <tag> scene spp w h dir1 dir2 ...   (time the same render with several builds of the libraries)
TAG=$1; SCENE=$2; SPP=$3; W=$4; H=$5; shift 5
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
for rep in 1 2; do
for d in "$@"; do
  echo "== $d" >> gpurun_out/$TAG/ab.txt
  MI355RT_LIB_DIR=$GRAFT_REPO_ROOT/$d timeout -k 10 120 python -u tools/perf_sweep.py $SCENE $SPP $W $H $AB_SWEEP >> gpurun_out/$TAG/ab.txt 2>&1 || exit 1
done
done
cat gpurun_out/$TAG/ab.txt
