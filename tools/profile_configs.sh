# usage: GIT_HEAD=<commit> bash tools/profile_configs.sh <round-tag> [config ...]   (on the GPU box: work/time/roofline and rocprofv3 summaries per configuration)
set -e
R=${1:-r02}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$R
run() {  # name scene spp w h
  echo "== $1: roofline"; timeout -k 10 300 python3 -u tools/config_roofline.py $2 $3 $4 $5 gpurun_out/$R/${R}_$1_roofline.json > gpurun_out/$R/$1_roofline.log 2>&1
  echo "== $1: rocprofv3"; bash tools/profile.sh ${R}_$1 $2 $3 $4 $5
  python3 tools/prof_summary.py gpurun_out/prof_${R}_$1 > gpurun_out/$R/${R}_$1_rocprof.txt
  python3 tools/derive_counters.py gpurun_out/$R/${R}_$1_rocprof.txt gpurun_out/$R/${R}_$1_roofline.json "$GIT_HEAD" > gpurun_out/$R/$1_derived.log 2>&1
  rm -rf gpurun_out/prof_${R}_$1
}
# one pass of the default pass size each (2^30 paths)
shift || true
WANT=${*:-book_cover cornell book2 cover1m}
for c in $WANT; do
  case $c in
    book_cover) run book_cover book_cover 512 1920 1080 ;;
    cornell) run cornell cornell 512 1024 1024 ;;
    book2) run book2 book2 256 1920 1080 ;;
    cover1m) run cover1m cover1m 64 3840 2160 ;;
  esac
done
ls -la gpurun_out/$R
