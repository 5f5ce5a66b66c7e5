# usage (GPU box): bash tools/share_sweep.sh   -- what one member's share of a 1024-spp book_cover frame costs at N = 1, 2, 4, 8
cd $GRAFT_REPO_ROOT
for spp in 1024 512 256 128; do
  echo "== $spp spp"
  timeout -k 10 200 python3 tools/perf_sweep.py book_cover $spp 1920 1080 TIMING=1,0 || exit 1
done
