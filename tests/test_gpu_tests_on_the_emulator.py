"""The -m gpu tests, UNCHANGED, against the whole-library CPU emulation: tests/emul_torch.py swaps tools/emul/lib_emul's shared library (the product's
own kernel and host sources behind the same C-ABI) in for the GPU library and backs "cuda" tensors with shared mappings, so ggml_amd/ops.py, the
ctypes binding, the test logic and every kernel on the route run on the CPU.  Here: a selection that finishes in under a minute (prefill GEMM and FLASH_ATTN_EXT cases are in the long run below; their
kernels and host code are covered by test_build_static.py's whole-library and fattn harness tests).  The whole of
tests/test_gpu_widening.py minus the full-size shapes and the tests that need the plug-in (190 tests) takes ~25 minutes:

    CDNA4_TESTS_ON_EMULATOR=1 python -m pytest tests/test_gpu_widening.py -m gpu -q \\
        -k "not stock_harness and not gpt2 and not 4096-4096 and not 32768 and not 8192 and not kw7 and not kw12"

(Nothing here is a CPU path of the product: the emulated library exists under build/ for tests only, and ggml_amd/ cannot load it.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECTION = ("(test_q4_1_q5_1_iq4_nl_gemv_parity and (16-256-1 or 20-544-7 or 48-1024-8)) or test_two_part_gemm_needs_whole_panels or test_iq4_to_float_is_bit_exact "
             "or (test_q4_1_q5_1_iq4_nl_mul_mat_id and (4-1-False-1 or 8-2-False-1)) or (test_decode_with_k_not_a_multiple_of_64 and 544) "
             "or (test_cpy_f32_to_q4_1_q5_0_q5_1_is_byte_exact and uniform) or (test_weight_reencoding_is_exact and 9-512) or (test_q4_1_q5_1_iq4_nl_prefill_gemm and 16-256-9)")


def test_selected_gpu_tests_pass_on_the_emulator():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the -m gpu tests run on it, the emulation is refused (tests/emul_torch.py)")
    env = dict(os.environ, CDNA4_TESTS_ON_EMULATOR="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_widening.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", SELECTION],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    if "cannot host the emulation" in tail:
        pytest.skip("the environment cannot host the emulation")
    assert r.returncode == 0, tail
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 30, tail


# round 3's routes (int8 matrix-core kernel for the five formats, Q5_0 / IQ4_NL following their Q8_0 target onto it, BF16 and directly read Q8_0 / Q4_0
# K / V under FLASH_ATTN_EXT): the small cases of their GPU tests — each asserts bit-identity between two routes, which is host logic as much as kernel
SELECTION_R3 = [("test_gpu_parity.py", "test_small_batches_on_the_int8_matrix_cores and (37-768-9 or 100-512-5)", 10),
                ("test_gpu_widening.py", "(test_more_formats_prefill_gemm and 16-256-9) or test_iq4_nl_reencoding_is_exact_and_public "
                                         "or (test_flash_attn_ext_bf16_kv and 64-35) or (test_flash_attn_ext_quantized_kv and 64-35 and q8_0)", 5)]


# round 6: the pipelined FLASH_ATTN_EXT prefill kernel (LDS-DMA staging with source-side swizzles, the emulated LDS transpose read, scores a chunk ahead): no mask / ragged
# key count / fewer than 64 keys / exactly one chunk, each at 8 / 4 / 2 waves per work-group
SELECTION_R3.append(("test_gpu_widening.py", "test_flash_attn_ext_pipelined_kernel and (kw1 or kw5 or kw7 or kw8)", 4))
# ... and its chunk list: causal masks with the -inf chunks skipped (k_fa_mask_flags) against walking them, bit for bit (plain, and the general mode with softcap)
SELECTION_R3.append(("test_gpu_widening.py", "test_flash_attn_ext_skips_masked_chunks and (kw0 or kw2)", 2))
# ... its key split (three forced splits + the merge kernel; -inf chunks inside a split) and the grouped-query decode tile of k_flash_attn_split (heads of a K / V head as tile rows)
SELECTION_R3.append(("test_gpu_widening.py", "(test_flash_attn_ext_pipelined_kernel_with_a_key_split and kw1) or (test_flash_attn_ext_grouped_query_decode_shares_the_tile and (kw1 or kw2))", 3))
# round 5: resident kernel-native images — the registry, the verified build, the lookup of row slices and the bit-identity of the routes that use an image are host logic
# as much as kernels
SELECTION_R3.append(("test_gpu_resident.py", "(test_resident_image_serves_prefill and q5_0) or (test_dequantize_row_of_the_image and q3_K)", 2))
# Q4_0 on Q4_K's kernels through a resident Q4_0R image: k_gemm_kq_t64 128-row tiles (and the tail in its store); with EMU_CUS=4 the routes a 256-CU part takes at full size —
# k_gemm_r8 (whole rounds of 256 x 256 tiles) and k_gemm_kq_t64's 256-row tiles
SELECTION_R3.append(("test_gpu_resident.py", "(test_q4_0_resident_image_puts and (512-1024-96 or 300-768)) or (test_q4_0_resident_image_carries and 768-512)", 4))
SELECTION_R3.append(("test_gpu_resident.py", "test_q4_0_resident_image_puts and (1024-512-256 or 1280-512-200)", 2, {"EMU_CUS": "4"}))
# Q8_0 on k_gemm_r8 through a resident Q8_0R image: whole rounds, one ragged round, the co-resident split in two (EMU_CUS=4), and the unchanged small-grid route
SELECTION_R3.append(("test_gpu_resident.py", "test_q8_0_and_q6_K_resident_images_put and (512-1024-256 or (14 and 768-512-200))", 3, {"EMU_CUS": "4"}))
# grouped MUL_MAT_ID on Q4_0 experts through a resident image of the expert stack: k_gemm_kq_t64<Q4_0R, 128, IDS> with the plan's tile order
SELECTION_R3.append(("test_gpu_resident.py", "test_q4_0_expert_stack and 128-512", 1))
# the hand-off of quantized activations: the second product on the first one's image (C-ABI: act_image_key, mul_mat_prepared[_fused]) is bit-identical to quantizing again
SELECTION_R3.append(("test_gpu_act_share.py", "(test_second_product and 16) or test_prepared_fused_refuses or (test_norm_that_also and (256-512 or 2304-768) and 1-gain) or (test_the_image_a_norm_leaves and (800-5 or 96-3 or 768-9))", 10))
# round 6: several one-row products of one activation row in ONE launch (ggml_cdna4_mul_mat_group / k_gemv_q_fused_grp) equal the single calls bit for bit, four ragged matrices
# with and without bias, Q4_K / Q4_0 / Q6_K; what has no grouped form is refused with -2
SELECTION_R3.append(("test_gpu_group.py", "(ms2 and (12 or 2 or 14)) or refuses", 6))
# ... and the second MUL_MAT_ID of one (b, ids) on the first one's front (ggml_cdna4_mul_mat_id_prepared): bit-identical to the full call; other routes leave no front
SELECTION_R3.append(("test_gpu_moe_front.py", "4-2-True-96 or 8-2-False-40 or other_routes", 3))


def _merged(selections):
    """one pytest process per (test file, environment): the selections of a group joined with `or`, their minimum counts added (a process start costs ~10 s)"""
    groups = {}
    for s_ in selections:
        fname, sel, at_least, env = s_ if len(s_) == 4 else s_ + ({},)
        key = (fname, tuple(sorted(env.items())))
        g = groups.setdefault(key, [fname, [], 0, env])
        g[1].append("(" + sel + ")"); g[2] += at_least
    return [(g[0], " or ".join(g[1]), g[2], g[3]) for g in groups.values()]


@pytest.mark.parametrize("fname,sel,at_least,extra_env", _merged(SELECTION_R3), ids=lambda v: v if isinstance(v, str) and v.endswith(".py") else None)
def test_round3_routes_pass_on_the_emulator(fname, sel, at_least, extra_env):
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the -m gpu tests run on it, the emulation is refused (tests/emul_torch.py)")
    env = dict(os.environ, CDNA4_TESTS_ON_EMULATOR="1", **extra_env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", fname), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", sel],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    if "cannot host the emulation" in tail:
        pytest.skip("the environment cannot host the emulation")
    assert r.returncode == 0, tail
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= at_least, tail
