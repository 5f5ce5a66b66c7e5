# usage (GPU box): GIT_HEAD=<commit> bash tools/evidence.sh <round-tag>
# Everything a round's numbers come from, on one box: every GPU test, the per-configuration profiles
# (work counts, kernel trace, counter passes, derived figures), then bench.py for every configuration
# (which quotes the profiles' traffic only if they are of the same device sources).
R=${1:-r03}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$R
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 400 --timeout-method thread > gpurun_out/$R/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/$R/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/$R/pytest_gpu.log
bash tools/profile_configs.sh $R || exit 1
mkdir -p profiles && cp gpurun_out/$R/${R}_*_roofline.json gpurun_out/$R/${R}_*_rocprof.txt profiles/
bash tools/bench_all.sh $R material_demo book_cover cornell book2 cover1m
