#!/bin/bash
# round 4, session 7: k_gemm_r8 (32 x 256 wave tiles, in-register unpack) beside t64 / w4; exact mode, new quantizer, IQ4_XS stability on hardware
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s7; mkdir -p $O
/opt/rocm/bin/rocminfo 2>/dev/null | grep -i -m4 "xnack\|gfx950" > $O/rocminfo.txt
L=268435456; W=$((L+33554432)); R=$((L+67108864))
cd tools/microbench
for shape in "32768 8192 512" "16384 8192 512" "16384 4096 512" "4096 4096 2048" "8192 8192 1024" "32768 4096 512" "4096 4096 512" "4096 11008 512"; do
  GB_VARIANTS="0,$W,$R" GB_SPLITKS=0 GB_ROUNDS=5 timeout 120 ./gemm_bench $shape
done > ../../$O/gemm_bench.txt 2>&1
V=""; for a in 0 1 2 3 4 8 16 32 15; do V="$V,$((R + a*65536))"; done
GB_VARIANTS="${V:1}" GB_SPLITKS=0 GB_ROUNDS=3 timeout 200 ./gemm_bench_abl 32768 8192 512 > ../../$O/gemm_bench_abl.txt 2>&1
cd ../..
timeout 900 python -m pytest tests/test_gpu_exact.py -x -q -m gpu > $O/t_exact.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quantiz or q8_K or activation" > $O/t_quant.log 2>&1
timeout 900 python -m pytest tests/test_gpu_widening.py -x -q -m gpu -k "two_part or iq4 or prefill_gemm" > $O/t_iq4.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_gpt2.py -x -q -m gpu -k "reference_order or logits_vs_cpu or per_op" > $O/t_gpt2.log 2>&1
cp gpurun_out/gpt2_parity.jsonl $O/ 2>/dev/null; cp gpurun_out/gpt2_resync_exact_8.log $O/ 2>/dev/null
timeout 300 python bench.py --steps 200 --no-extras --no-cpu-baseline > $O/bench_lean.json 2> $O/bench_lean.err
tail -3 $O/t_exact.log $O/t_quant.log $O/t_iq4.log $O/t_gpt2.log; cat $O/rocminfo.txt; cat $O/gemm_bench.txt | grep -v "^$" | cut -c1-150; cat $O/gemm_bench_abl.txt | cut -c1-120; cat $O/bench_lean.json | cut -c1-600
