set -e
B=$PWD/vitpose_amd/lib/ab/libvitpose_hip_builtin.so
timeout -k 10 600 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_kernels_gpu.py 2>&1 | tail -3
for lib in asm builtin; do
  if [ $lib = builtin ]; then export VITPOSE_HIP_LIB=$B; else unset VITPOSE_HIP_LIB; fi
  echo "== $lib"
  timeout -k 10 120 python tools/gemm_bench.py --tile-mode 3 --only "dgrad" 2>&1 | grep -v amdgpu.ids
  timeout -k 10 120 python tools/gemm_bench.py --tile-mode 4 --only "DGELU" 2>&1 | grep -v amdgpu.ids
  timeout -k 10 120 python tools/gemm_bench.py --tile-mode 3 --only "wgrad_group" 2>&1 | grep -v amdgpu.ids
  timeout -k 10 120 python tools/gemm_bench.py --tile-mode 2 --only "wgrad_group" 2>&1 | grep -v amdgpu.ids
done
unset VITPOSE_HIP_LIB
for m in 3 4 5; do
  echo "== mode $m"
  timeout -k 10 120 python tools/gemm_bench.py --tile-mode $m --only "fwd_qkv" 2>&1 | grep -v amdgpu.ids
  timeout -k 10 120 python tools/gemm_bench.py --tile-mode $m --only "GELU" 2>&1 | grep -v amdgpu.ids
done
for lib in asm builtin asm builtin; do
  if [ $lib = builtin ]; then export VITPOSE_HIP_LIB=$B; else unset VITPOSE_HIP_LIB; fi
  echo "== bench $lib"
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | grep -o '"value": [0-9.]*, \|"ms_per_step": [0-9.]*\|"step_ms_median": [0-9.]*' | tr '\n' ' '; echo
done
