#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/ablate.txt
export PYTHONUNBUFFERED=1
for ab in 0 1 2 4 8 16 3 6 12 14 15 31 30; do
  CDNA4_GEMM_ABLATE=$ab timeout -k 10 120 python bench.py --steps 100 --warmup 10 --variant 23 --splitk 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('ablate',$ab,'gemm_us',j['roofline']['us_per_launch'])" >> gpurun_out/ablate.txt 2>&1
done
cat gpurun_out/ablate.txt
