for seed in 11 12 13 14 15 16; do
  FUZZ_BIG=1 FUZZ_SEED=$seed FUZZ_ITERS=60 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shape_fuzz_against_oracle" 2>&1 | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl\|^Some\|0\.[0-9][0-9] \|deltas = " | tail -3
done
