# usage: bash tools/profile.sh <tag> [scene] [spp]   (run on the GPU box through gpurun)
set -e
TAG=${1:-r01}; SCENE=${2:-book_cover}; SPP=${3:-32}; W=$4; H=$5
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 tools/profile_run.py $SCENE $SPP $W $H > $OUT/trace.log 2>&1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
            "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 200 rocprofv3 --pmc $pass -d $OUT/pmc_$n -o pmc -- python3 tools/profile_run.py $SCENE $SPP $W $H > $OUT/pmc_$n.log 2>&1 || echo "pass $n failed" >> $OUT/failed.txt
done
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
