"""The ggml plug-in's HOST logic without a GPU: ggml_amd/csrc/backend/*.cpp built a second time against a host stand-in for the HIP runtime (tools/emul/shim_plugin) and the
whole-library CPU emulation of the kernel library, driven through ggml's PUBLIC API by oracle/_ref/split_harness (the unmodified reference's libggml-base / ggml-cpu beside it) —
tools/emul/plugin_emul_check.py.  What a GPU session checks at full size, at sizes the emulation finishes in seconds: the graph walk and its peepholes, the hand-off of
quantized activations between MUL_MATs of one src1, NORM chains that leave the activation image (Q8_K and Q8_0 classes), the resident buffer type and its re-layouts.
(Test infrastructure: nothing here is a CPU path of the product — the emulated libraries live under build/ and ggml_amd/ cannot load them.)"""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def plug():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the -m gpu tests drive the real plug-in on it")
    spec = importlib.util.spec_from_file_location("plugin_emul_check", os.path.join(ROOT, "tools", "emul", "plugin_emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not mod.available():
        pytest.skip("needs the reference tree (ggml headers), oracle/_ref/split_harness and ROCm's clang")
    return mod


def _shared(plug, type_, d, h, b, share):
    env = {"HARNESS_NO_TIMING": 1}
    if not share:
        env["GGML_CDNA4_NO_ACT_SHARE"] = 1
    j = plug.harness([type_, d, h, b, "shared"], env=env)
    if j is None:
        pytest.skip("the environment cannot host the emulation")
    return j


@pytest.mark.parametrize("type_,d,h,b,hand_offs", [("q4_K", 256, 512, 96, 5), ("q4_0", 256, 512, 96, 5), ("q8_0", 256, 512, 16, 3), ("q4_K", 256, 512, 1, 0)])
def test_layer_front_through_ggmls_public_api_on_the_emulated_plugin(plug, type_, d, h, b, hand_offs):
    """rms_norm -> {wq, wk, wv + bias}, rms_norm -> {w_gate, w_up} -> w_down (oracle/split_harness.cpp `shared`): the NORM chains leave the activation image on the fp16 GEMM routes
    (Q8_K image for K-quants, Q8_0 image for Q4_0 — the k_norm<.., 2> path that was written after the round's GPU time ran out), wk / wv / w_up multiply the previous product's
    image, the int8 routes of 16 rows share without a producer, one row shares nothing; the outputs' bytes equal those of a run with the hand-off off; K and V within the
    1e-3 bar of the reference CPU backend"""
    on, off = _shared(plug, type_, d, h, b, True), _shared(plug, type_, d, h, b, False)
    assert on["act_hand_offs_first_compute"] == hand_offs and off["act_hand_offs_first_compute"] == 0, (on, off)
    assert on["fnv1a"] == off["fnv1a"], (on, off)
    assert on["k_vs_cpu"] < 1e-3 and on["v_vs_cpu"] < 1e-3, on            # (`out`, two re-quantizations deep, is reported by the harness and not bounded: see the reference-order test below)


@pytest.mark.parametrize("type_,b", [("q4_K", 9), ("q5_K", 5), ("q6_K", 12), ("q4_0", 7), ("q4_K", 1)])
def test_layer_front_in_reference_order_is_the_cpu_backends_bits_on_the_emulated_plugin(plug, type_, b):
    """GGML_CDNA4_EXACT=1: RMS_NORM, six quantized MUL_MATs — K-quants in the AVX2 lane order of ggml_vec_dot_q4_K_q8_K / _q5_K_q8_K / _q6_K_q8_K
    (ggml_amd/csrc/exact.hip: k_mul_mat_exact_kq) —, SILU, MUL, ADD: every fp32 word of K, V and out equals the reference CPU backend's, three products and two re-quantizations deep
    (tests/test_gpu_act_share.py runs the same on the hardware at full width)"""
    j = plug.harness([type_, 256, 512, b, "shared"], env={"HARNESS_NO_TIMING": 1, "GGML_CDNA4_EXACT": 1})
    if j is None:
        pytest.skip("the environment cannot host the emulation")
    assert j["act_hand_offs_first_compute"] == 0 and j["grouped_first_compute"] == 0, j
    assert j["words_differing_from_cpu"] == 0 and j["out_vs_cpu"] == 0.0, j


def test_moe_ffn_second_stack_multiplies_the_first_ones_front_on_the_emulated_plugin(plug):
    """round 6: a mixture-of-experts FFN as llama.cpp's build_moe_ffn issues it (oracle/split_harness.cpp `moeffn`: up and gate = MUL_MAT_ID of the same (cur, ids), down behind
    silu(gate) * up): the gate stack multiplies the front the up stack's call left in the workspace — sorted ids, tile records, spans, quantized activations; one launch instead of
    two — and every output byte equals the run with the sharing off; the down stack (other activations) makes its own front"""
    on = plug.harness(["q4_K", 256, 512, 96, "moeffn"], env={"HARNESS_NO_TIMING": 1})
    off = plug.harness(["q4_K", 256, 512, 96, "moeffn"], env={"HARNESS_NO_TIMING": 1, "GGML_CDNA4_NO_ACT_SHARE": 1})
    if on is None or off is None:
        pytest.skip("the environment cannot host the emulation")
    assert on["moe_fronts_shared_first_compute"] == 1 and off["moe_fronts_shared_first_compute"] == 0, (on, off)
    assert on["fnv1a"] == off["fnv1a"], (on, off)
    assert on["out_vs_cpu"] < 2e-2, on                                    # (sanity only: two grouped fp16 GEMMs and a re-quantization deep)


@pytest.mark.parametrize("type_", ["q4_K", "q4_0"])
def test_one_row_products_of_one_src1_run_as_one_launch_on_the_emulated_plugin(plug, type_):
    """round 6: at ONE activation row the layer front's products of `cur` (Wk, Wv + its bias, Wq + the residual row, which at one row is a bias) and of `f`
    (w_gate, w_up) are two grouped launches instead of five (ggml_cdna4_mul_mat_group / k_gemv_q_fused_grp; the plug-in runs Wq and w_up EARLY, with the first product of their src1, after checking that
    nothing in between touches their outputs' memory): three products rode along, and every output byte equals the node-by-node run (GGML_CDNA4_NO_GROUP=1)"""
    on = plug.harness([type_, 256, 512, 1, "shared"], env={"HARNESS_NO_TIMING": 1})
    off = plug.harness([type_, 256, 512, 1, "shared"], env={"HARNESS_NO_TIMING": 1, "GGML_CDNA4_NO_GROUP": 1})
    if on is None or off is None:
        pytest.skip("the environment cannot host the emulation")
    assert on["grouped_first_compute"] == 3 and off["grouped_first_compute"] == 0, (on, off)
    assert on["fnv1a"] == off["fnv1a"], (on, off)
    assert on["k_vs_cpu"] < 1e-3 and on["v_vs_cpu"] < 1e-3, on


@pytest.mark.parametrize("type_,m,k,b", [("q4_0", 256, 512, 96), ("q8_0", 256, 512, 40), ("q6_K", 128, 512, 96), ("q5_0", 128, 256, 40), ("q4_K", 128, 256, 40)])
def test_resident_buffer_type_on_the_emulated_plugin(plug, type_, m, k, b):
    """the CDNA4_Resident extra buffer type (ggml_backend_dev_get_extra_bufts): weights through ggml_backend_tensor_set, MUL_MAT at prefill and decode sizes, a rewrite — against the
    default buffer type and the CPU backend (tests/test_gpu_resident.py's plug-in test, at emulation sizes; on a 256-CU part these small grids keep their per-call kernels, so the
    image must simply not disturb anything: bit-identical)"""
    j = plug.harness([type_, m, k, b, "resident"], env={"HARNESS_NO_TIMING": 1})
    if j is None:
        pytest.skip("the environment cannot host the emulation")
    assert j["buft"].startswith("CDNA4_Resident") and j["set_get_roundtrip"] is True
    assert j["decode_bit_identical_to_default"] is True
    assert j["resident_vs_default_rel_l2"] < 1e-5 and j["rewritten_vs_default_rel_l2"] < 1e-5 and j["resident_vs_cpu_rel_l2"] < 1e-3, j


@pytest.mark.skipif(not os.environ.get("CDNA4_FULL_CPU_SUITE"), reason="35 s: the C-ABI selection of tests/test_gpu_tests_on_the_emulator.py covers k_gemm_r8<Q8_0R>; CDNA4_FULL_CPU_SUITE=1 runs it through the plug-in too")
def test_resident_q8_0_takes_k_gemm_r8_on_a_small_pretend_chip(plug):
    """EMU_CUS=4: a 512 x 1024 x 130 Q8_0 product is a quarter .. half tile per CU — with the image the plug-in's MUL_MAT runs k_gemm_r8<Q8_0R> (co-resident split in two), without it the
    staging kernel: same result to 1e-5, within the bar of the CPU backend"""
    j = plug.harness(["q8_0", 512, 1024, 130, "resident"], env={"HARNESS_NO_TIMING": 1, "EMU_CUS": 4})
    if j is None:
        pytest.skip("the environment cannot host the emulation")
    assert j["resident_bit_identical_to_default"] is False              # (another kernel, another summation order)
    assert j["resident_vs_default_rel_l2"] < 1e-5 and j["resident_vs_cpu_rel_l2"] < 1e-3 and j["decode_bit_identical_to_default"] is True, j


@pytest.mark.parametrize("type_,m,k,tokens,other_kernel", [("q4_0", 128, 512, 100, True), ("q4_K", 128, 256, 70, False)])
def test_expert_stack_in_the_resident_buffer_type_on_the_emulated_plugin(plug, type_, m, k, tokens, other_kernel):
    """MUL_MAT_ID through ggml's public API (oracle/split_harness.cpp `moe`: a 3-D expert tensor, 4 experts, 2 used): a Q4_0 stack in the resident buffer type gets ONE image and
    prefill-sized calls run Q4_K's grouped kernel on it (another kernel than the default buffer type's: not bit-identical, 1e-5 apart); a Q4_K stack needs no image (bit-identical);
    a single token reads the source bytes either way"""
    j = plug.harness([type_, m, k, tokens, "moe"], env={"HARNESS_NO_TIMING": 1})
    if j is None:
        pytest.skip("the environment cannot host the emulation")
    assert j["resident_vs_cpu_rel_l2"] < 1e-3 and j["default_vs_cpu_rel_l2"] < 1e-3 and j["resident_vs_default_rel_l2"] < 1e-5, j
    assert j["resident_bit_identical_to_default"] is (not other_kernel), j
    assert j["one_token_bit_identical_to_default"] is True and j["one_token_vs_cpu_rel_l2"] < 1e-5, j
