#!/bin/bash
# round 4, session 1: k_gemm_lds against k_gemm_kq_t64 on one box (gemm_bench, interleaved rounds), its ablations + phase trace, and the IQ4_XS diagnosis
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s1; mkdir -p $O
L=268435456; L128=$((L+536870912)); L256=$((L+1073741824)); T256=$((8192+32768)); T128=$((8192+16384))
cd tools/microbench
for shape in "32768 8192 512" "16384 8192 512" "16384 4096 512" "8192 8192 512" "4096 11008 512" "4096 4096 512" "4096 14336 512"; do
  GB_VARIANTS="0,$L" GB_SPLITKS=0 GB_ROUNDS=5 timeout 120 ./gemm_bench $shape
done > ../../$O/gemm_bench.txt 2>&1
{ GB_VARIANTS="$T256,$L256,$L128" GB_SPLITKS=0 GB_ROUNDS=4 timeout 120 ./gemm_bench 32768 8192 512
  GB_VARIANTS="0,$L128,$L256" GB_SPLITKS=0 GB_ROUNDS=4 timeout 120 ./gemm_bench 4096 4096 512
  GB_VARIANTS="0,$L128" GB_SPLITKS=2 GB_ROUNDS=4 timeout 120 ./gemm_bench 4096 4096 512
  GB_VARIANTS="0,$L128,$L256" GB_SPLITKS=0 GB_ROUNDS=4 timeout 120 ./gemm_bench 4096 11008 512
  GB_VARIANTS="0,$L128,$L256" GB_SPLITKS=0 GB_ROUNDS=4 timeout 120 ./gemm_bench 4096 4096 1024
  GB_VARIANTS="0,$L128,$L256" GB_SPLITKS=0 GB_ROUNDS=4 timeout 120 ./gemm_bench 4096 4096 256
} > ../../$O/gemm_bench_forced.txt 2>&1
# ablations of the 256-row form at C5 and of the 128-row form at 16384x8192x512 (timing only), then the phase trace
A() { echo $(( $1 + ($2 << 16) )); }
{ GB_VARIANTS="$L256,$(A $L256 1),$(A $L256 2),$(A $L256 3),$(A $L256 4),$(A $L256 8),$(A $L256 16),$(A $L256 32),$(A $L256 64),$(A $L256 15)" GB_SPLITKS=1 GB_ROUNDS=3 timeout 300 ./gemm_bench_abl 32768 8192 512 "$(A $L256 256)"
  GB_VARIANTS="$L128,$(A $L128 1),$(A $L128 2),$(A $L128 3),$(A $L128 4),$(A $L128 8),$(A $L128 16),$(A $L128 32),$(A $L128 64),$(A $L128 15)" GB_SPLITKS=1 GB_ROUNDS=3 timeout 300 ./gemm_bench_abl 16384 8192 512 "$(A $L128 256)"
  GB_TRACE_REPS=20 GB_VARIANTS="$L256" GB_SPLITKS=1 GB_ROUNDS=1 timeout 300 ./gemm_bench_abl 32768 8192 512 "$(A $L256 256)"
} > ../../$O/gemm_bench_abl.txt 2>&1
cd ../..
CDNA4_IQ4_XS_GEMM=1 CDNA4_DIAG_CONVERT_ANY=1 timeout 600 python scripts/gpu_diag_iq4xs2.py > $O/iq4xs_diag2.log 2>&1
CDNA4_IQ4_XS_GEMM=1 CDNA4_DIAG_CONVERT_ANY=1 timeout 600 python scripts/gpu_diag_iq4xs2.py Q2_K > $O/q2k_diag2.log 2>&1
tail -5 $O/iq4xs_diag2.log
head -40 $O/gemm_bench.txt
