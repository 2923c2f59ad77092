#!/bin/bash
# round 4, session 5: k_gemm_w4 (one wave per SIMD, 128 x 128 wave tiles) against k_gemm_kq_t64 and k_gemm_lds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
L=268435456; W=$((L+33554432)); W128=$((W+536870912)); W256=$((W+1073741824))
cd tools/microbench
for shape in "32768 8192 512" "16384 8192 512" "16384 4096 512" "8192 8192 512" "4096 11008 512" "4096 4096 512" "4096 14336 512" "4096 4096 1024" "4096 4096 2048"; do
  GB_VARIANTS="0,$L,$W" GB_SPLITKS=0 GB_ROUNDS=5 timeout 120 ./gemm_bench $shape
done > ../../$O/gemm_bench.txt 2>&1
{ GB_VARIANTS="0,$W128,$W256" GB_SPLITKS=0 GB_ROUNDS=4 timeout 120 ./gemm_bench 32768 8192 512
  GB_VARIANTS="0,$W128,$W256" GB_SPLITKS=0 GB_ROUNDS=4 timeout 120 ./gemm_bench 4096 4096 512
  GB_VARIANTS="0,$W128,$W256" GB_SPLITKS=0 GB_ROUNDS=4 timeout 120 ./gemm_bench 4096 11008 512
  GB_VARIANTS="0,$W128" GB_SPLITKS=2 GB_ROUNDS=4 timeout 120 ./gemm_bench 4096 4096 512
  GB_VARIANTS="0,$W128,$W256" GB_SPLITKS=0 GB_ROUNDS=4 timeout 120 ./gemm_bench 8192 8192 512
} > ../../$O/gemm_bench_forced.txt 2>&1
A() { echo $(( $1 + ($2 << 16) )); }
{ GB_VARIANTS="$W256,$(A $W256 1),$(A $W256 2),$(A $W256 3),$(A $W256 4),$(A $W256 8),$(A $W256 16),$(A $W256 32),$(A $W256 15)" GB_SPLITKS=1 GB_ROUNDS=3 timeout 300 ./gemm_bench_abl 32768 8192 512
  GB_VARIANTS="$W128,$(A $W128 1),$(A $W128 2),$(A $W128 3),$(A $W128 4),$(A $W128 8),$(A $W128 16),$(A $W128 32),$(A $W128 15)" GB_SPLITKS=1 GB_ROUNDS=3 timeout 300 ./gemm_bench_abl 16384 8192 512
} > ../../$O/gemm_bench_abl.txt 2>&1
cd ../..
cat $O/gemm_bench.txt; cat $O/gemm_bench_forced.txt
