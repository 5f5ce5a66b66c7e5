set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx" | head -4 > gpurun_out/rocminfo.txt 2>&1 || true
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py -x -v --timeout 240 --timeout-method thread -m gpu > gpurun_out/pytest_gpu1.log 2>&1
