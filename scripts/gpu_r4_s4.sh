#!/bin/bash
# round 4, session 3: k_gemm_lds v3 (producer arithmetic + fragment reads in the MFMA gaps) against k_gemm_kq_t64; phase traces; IQ4_XS: variants 0 / 8 / shipped, in-library route stability
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
L=268435456; L128=$((L+536870912)); L256=$((L+1073741824))
cd tools/microbench
for shape in "32768 8192 512" "16384 8192 512" "16384 4096 512" "8192 8192 512" "4096 11008 512" "4096 4096 512" "4096 14336 512"; do
  GB_VARIANTS="0,$L" GB_SPLITKS=0 GB_ROUNDS=5 timeout 120 ./gemm_bench $shape
done > ../../$O/gemm_bench.txt 2>&1
A() { echo $(( $1 + ($2 << 16) )); }
{ GB_VARIANTS="$L256,$(A $L256 3),$(A $L256 4),$(A $L256 8),$(A $L256 16),$(A $L256 32),$(A $L256 64),$(A $L256 15)" GB_SPLITKS=1 GB_ROUNDS=3 timeout 300 ./gemm_bench_abl 32768 8192 512 "$(A $L256 256)"
  GB_VARIANTS="$L128,$(A $L128 1),$(A $L128 3),$(A $L128 4),$(A $L128 8),$(A $L128 16),$(A $L128 32),$(A $L128 64),$(A $L128 15)" GB_SPLITKS=1 GB_ROUNDS=3 timeout 300 ./gemm_bench_abl 16384 8192 512 "$(A $L128 256)"
} > ../../$O/gemm_bench_abl.txt 2>&1
cd ../..
cat $O/gemm_bench.txt
