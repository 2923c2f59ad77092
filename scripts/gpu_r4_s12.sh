#!/bin/bash
# round 4, session 12: t64 128- vs 256-row tiles at one tile per CU (same box); the new parity tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/s12; mkdir -p $O
( cd tools/microbench
  for shape in "16384 8192 512" "16384 4096 512" "4096 4096 2048" "8192 8192 1024" "8192 4096 1024" "16384 11008 512"; do GB_SPLITKS=0 GB_VARIANTS="24583,40967,0" GB_ROUNDS=5 timeout 180 ./gemm_bench $shape ""; done
) 2>&1 | grep -E "^M=|^variant" > $O/t64_tiles.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cabi_ops.py -q -m gpu --tb=short -k "few_rows_per_expert or full_matrix or reference_quantized_rows or mul_mat_id or shared_device or large_grid or auto_picks" > $O/t_new.log 2>&1
cat $O/t64_tiles.txt | cut -c1-130; tail -6 $O/t_new.log; grep -h "few_rows_per_expert\|headline_full_matrix\|c5_reference\|shared_device" gpurun_out/parity_report.jsonl | tail -12 | cut -c1-400
