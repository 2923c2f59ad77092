#!/bin/bash
# First GPU call of the next round (~1 minute of box time): the three measurements DESIGN.md §7 items 0 / 1(a) ask for.
# Needs `make -C tools/microbench all abl` done in the container (the binaries travel with the snapshot).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
cd tools/microbench
# (1) the matrix-pipe price of the unpack mix, the clock the chip holds, and the sustained MFMA-only rate (random / zero data)
timeout 120 ./mfma_valu 500 > ../../gpurun_out/mfma_valu.txt 2>&1
# (1b) the floor of a one-launch 9.4 MB stream (the decode GEMV's denominator): empty-kernel launch cost, pure-load kernels
timeout 120 ./launch_floor > ../../gpurun_out/launch_floor.txt 2>&1
# (2) the GPU fault of the 8192x8192x512 sweep: one process per kernel, no trace kernels (argv[4] empty), line-buffered output
{
for v in 4119 2101271 31461399 23 2071 663; do
    echo "== variant $v"
    GB_VARIANTS=$v GB_SPLITKS=1 GB_ROUNDS=2 timeout 60 ./gemm_bench_abl 8192 8192 512 "" 2>&1 | tail -3
    echo "   exit $?"
done
echo "== trace build of the first-cut kernel (variant 23) at this shape"
GB_VARIANTS=4119 GB_SPLITKS=1 GB_ROUNDS=1 timeout 60 ./gemm_bench_abl 8192 8192 512 23 2>&1 | tail -4
} > ../../gpurun_out/fault_8192.txt 2>&1
cd ../..
# (2b) candidates that must stay bit-identical (rel-L2 0 against the first config): early table read (EXP 1), balanced epilogue (EXP 2), stores straight from registers (EXP 4), both (EXP 6)
cd tools/microbench
{ GB_VARIANTS=4119,69655,135191,266263,397335 GB_SPLITKS=0,1 timeout 60 ./gemm_bench_abl 4096 4096 512 "" 2>&1 | tail -12
  GB_VARIANTS=4119,135191,266263,397335 GB_SPLITKS=1 timeout 60 ./gemm_bench_abl 8192 4096 512 "" 2>&1 | tail -6; } > ../../gpurun_out/w12_candidates.txt 2>&1
# (2d) the experimental 4 + 4-wave kernels (variant 8199: 256x128 tile / 4 compute waves, 24583: 128x128 tile, 40967: 256x128 / 8 compute waves) next to the default (rel-L2 must be ~1e-7): one shape per process,
#      a short timeout each — a barrier mismatch in a first run would hang
{ for shape in "1024 1024 256" "4096 4096 512" "8192 4096 512" "32768 8192 512"; do
    GB_VARIANTS=4119,8199,24583,40967 GB_SPLITKS=0 GB_ROUNDS=2 timeout 30 ./gemm_bench_abl $shape "" 2>&1 | tail -5; echo "   exit $?"
  done; } > ../../gpurun_out/x4l.txt 2>&1
# (2c) the clock the chip settles at while the shipped / no-DMA / stripped loops run back to back (EXP bit 512 records it)
{ for shape in "4096 4096" "8192 4096"; do
    GB_VARIANTS=4119 GB_SPLITKS=0 GB_ROUNDS=1 GB_TRACE_SPLITK=0 GB_TRACE_REPS=400 timeout 60 ./gemm_bench_abl $shape 512 33558551,35655703,65015959 2>&1 | grep "block 0"
  done; } > ../../gpurun_out/w12_clock.txt 2>&1
cd ../..
# (3) the default path at that shape against the slice-per-barrier kernel (variant 5) through the C-ABI
timeout 300 python - > gpurun_out/parity_8192.txt 2>&1 <<'PY'
import numpy as np, torch
from ggml_amd import native, ops
from bench import synth_q4k, Q4_K
L = native.lib(); dev = torch.device("cuda", 0)
M, K, B = 8192, 8192, 512
a = ops.QTensor.from_host_bytes(Q4_K, K, M, synth_q4k(M, K, 7), device=dev)
x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (B, K)).astype(np.float32)).to(dev)
ws = torch.empty(L.ggml_cdna4_mul_mat_workspace_size(Q4_K, K, B), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
out = {}
for variant in (0, 5):
    y = torch.empty((B, M), dtype=torch.float32, device=dev)
    native.check(L.ggml_cdna4_mul_mat(Q4_K, a.data.data_ptr(), a.row_bytes, x.data_ptr(), K, y.data_ptr(), M, M, K, B, ws.data_ptr(), ws.numel(), ops.PATH_GEMM, variant, 0, st))
    torch.cuda.synchronize(); out[variant] = y.double()
print("rel-L2 default vs variant 5 at 8192x8192x512:", float((out[0] - out[5]).norm() / out[5].norm()))
PY
# (4) the opt-in tests of code written without a GPU: GGUF upload, the 4 + 4-wave kernel
CDNA4_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gguf.py -q -m gpu -k upload > gpurun_out/experimental_tests.txt 2>&1
CDNA4_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "4plus4 or loader_wave_kernel_q5_k" >> gpurun_out/experimental_tests.txt 2>&1
# (5) the one-launch step: k_gemm_kq_w12<Q4_K> with the activation quantizer inside the launch (variant 4119 | 1024 << 16; verified on
#     the CPU emulator) — bit-identity with the default path incl. workspace reuse and > 64 launches, then the step time beside the default
CDNA4_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "in_launch" >> gpurun_out/experimental_tests.txt 2>&1
timeout 200 python bench.py --fuseq-leg --steps 200 > gpurun_out/fuseq_leg.txt 2>&1
# (5c) Q5_0 / Q2_K / Q3_K through the GEMV units (CDNA4_EXTRA_TYPES=1 in a child process) against the oracle
CDNA4_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "extra_weight_types" >> gpurun_out/experimental_tests.txt 2>&1
# (5b) the same quantizer on the AUTO route (CDNA4_FUSEQ=1: every Q4_K prefill GEMM whose auto kernel is k_gemm_kq_w12 becomes one launch): the whole
#      GEMM parity suite under it — what has to be green before the knob becomes the default
CDNA4_FUSEQ=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gemm or mul_mat or full_size" > gpurun_out/pytest_fuseq_auto.log 2>&1; tail -3 gpurun_out/pytest_fuseq_auto.log
tail -5 gpurun_out/experimental_tests.txt
cat gpurun_out/fuseq_leg.txt | tail -2
tail -2 gpurun_out/mfma_valu.txt; cat gpurun_out/launch_floor.txt; cut -c1-200 gpurun_out/w12_candidates.txt; cat gpurun_out/w12_clock.txt; cut -c1-200 gpurun_out/x4l.txt; cat gpurun_out/fault_8192.txt | cut -c1-200; cat gpurun_out/parity_8192.txt | tail -3
