# usage: bash tools/ab_multi.sh <tag> "dir1 dir2 ..." "scene:spp:w:h ..."   (GPU box: time each scene with each build of the libraries)
TAG=$1; LIBS=$2; CASES=$3
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
for c in $CASES; do
  IFS=: read SCENE SPP W H <<< "$c"
  for rep in 1 2; do
  for d in $LIBS; do
    echo "== $d $c" >> gpurun_out/$TAG/ab.txt
    MI355RT_LIB_DIR=$GRAFT_REPO_ROOT/$d timeout -k 10 120 python -u tools/perf_sweep.py $SCENE $SPP $W $H $AB_SWEEP >> gpurun_out/$TAG/ab.txt 2>&1 || exit 1
  done
  done
done
cat gpurun_out/$TAG/ab.txt
