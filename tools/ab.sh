#!/bin/bash
# dev: bench with alternative builds of libphmm_amd.so (tools/_ab/lib*.so)
for v in "$@"; do
  cp tools/_ab/lib$v.so dbgphmm_amd/libphmm_amd.so
  echo "== $v"
  timeout -k 10 120 python -m pytest tests/test_gpu_dense.py tests/test_gpu_sparse.py -m gpu -q -x 2>&1 | tail -1
  timeout -k 10 250 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.1f  bwd %.0f GB/s (%.0f us)  fwd %.0f GB/s (%.0f us)' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'], d['roofline']['fwd_step']['achieved'], d['roofline']['fwd_step']['avg_launch_us']))"
done
