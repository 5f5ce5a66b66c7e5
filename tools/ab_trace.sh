# usage: bash tools/ab_trace.sh <tag> scene spp w h dir1 dir2 ...   (per-kernel durations of the same render with several builds)
TAG=$1; SCENE=$2; SPP=$3; W=$4; H=$5; shift 5
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for d in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/$(basename $d)
  mkdir -p $OUT
  MI355RT_LIB_DIR=$GRAFT_REPO_ROOT/$d timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python3 tools/profile_run.py $SCENE $SPP $W $H > $OUT/run.log 2>&1 || exit 1
  echo "== $d"
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"][:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
for n, (c, t) in agg.items():
    if t > 0.05: print(f"  {n:60s} {c:4d} launches {t:9.3f} ms")
PY
done
