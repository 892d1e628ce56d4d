set -e
timeout -k 10 600 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_kernels_gpu.py -k "w2" 2>&1 | tail -2
for lib in new old new old; do
  if [ $lib = new ]; then unset VITPOSE_HIP_LIB; else export VITPOSE_HIP_LIB=$PWD/vitpose_amd/lib/ab/lib_old.so; fi
  echo "== $lib"
  timeout -k 10 120 python tools/gemm_bench.py --tile-mode 4 --only "GELU" 2>&1 | grep -v amdgpu.ids
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | grep -o '"value": [0-9.]*, \|"ms_per_step": [0-9.]*\|"step_ms_median": [0-9.]*' | tr '\n' ' '; echo
done
