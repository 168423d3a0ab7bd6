export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gemm_gpu.py -x -q -k frozen 2>&1 | tail -30
