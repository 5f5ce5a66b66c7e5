cd $GRAFT_REPO_ROOT
for c in "book_cover 8 1920 1080" "book_cover 16 1920 1080" "book_cover 32 1920 1080" "cornell 8 1024 1024" "cornell 16 1024 1024" "cornell 32 1024 1024" "material_demo 256 400 225" "book2 8 1920 1080" "book2 16 1920 1080"; do
  echo "== $c"
  timeout -k 10 200 python3 tools/perf_sweep.py $c MI355RT_STREAM=0,1,0,1 TIMING=0 || exit 1
done
