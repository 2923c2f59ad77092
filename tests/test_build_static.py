"""Static properties of the compiled gfx950 kernels that the performance of the hot path depends on, checked from the
assembly hipcc emits (no GPU needed): the shipped kernels must not spill to scratch, the 12-wave kernel must fit three waves
per SIMD, and no work-group may ask for more than the CU's 160 KB of LDS."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module", autouse=True)
def _emulators_built_in_parallel():
    """the CPU emulators (tools/emul) are built on demand by their check modules, one after the other; on a fresh checkout that is most of this
    file's run time, so build all of them here at once (each build is a no-op when its binary is newer than its sources)"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        return
    import importlib.util
    from concurrent.futures import ThreadPoolExecutor

    def mod(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "emul", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    jobs = [lambda: mod("emul_check").build("w12"), lambda: mod("emul_check").build("w8"), lambda: mod("emul_check").build("t64")]
    jobs += [mod(n).build for n in ("gemv_emul_check", "quant_emul_check", "convert_emul_check", "deq_emul_check", "fattn_emul_check", "lib_emul_check")]
    with ThreadPoolExecutor(max_workers=9) as ex:
        for f in [ex.submit(j) for j in jobs]:
            try:
                f.result()
            except Exception:  # noqa: BLE001 — the test that needs this emulator reports the build error
                pass


def _isa_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_manifest", os.path.join(ROOT, "tools", "isa_manifest.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def gemm_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    return _isa_tool().asm_of("gemm_q_mfma.hip")          # (cached compile shared with the manifest check: tools/isa_manifest.py)


def _prop(asm, kernel, name):                                        # (plain finds: the assembly of one source is tens of MB)
    key = ".set %s.%s, " % (kernel, name)
    i = asm.find(key)
    assert i >= 0, "kernel %s not found in the assembly" % kernel
    return int(re.match(r"\d+", asm[i + len(key):i + len(key) + 16]).group(0))


def _loops(body):
    """the depth-1 loops of a kernel body: from each loop header to the first s_cbranch_scc1 behind it (plain finds: a lazy DOTALL regex is quadratic on a chunk without one)"""
    out = []
    for chunk in body.split("Loop Header: Depth=1")[1:]:
        j = chunk.find("s_cbranch_scc1")
        if j >= 0:
            out.append(chunk[:j + len("s_cbranch_scc1")])
    return out


def _lds(asm, kernel):
    i = asm.find(".amdhsa_kernel %s\n" % kernel)
    assert i >= 0, kernel
    key = ".amdhsa_group_segment_fixed_size "
    j = asm.index(key, i)
    return int(re.match(r"\d+", asm[j + len(key):j + len(key) + 16]).group(0))


# mangled names: k_gemm_kq_w12<Q4_K, true, 0>, k_gemm_kq_w8p<Q5_K, false>, k_gemm_kq_w8<Q4_K, false, 20>
SHIPPED = [
    "_Z13k_gemm_kq_w12ILi12ELb1ELi0EEv11gemm_params",
    "_Z13k_gemm_kq_w8pILi13ELb0EEv11gemm_params",
    "_Z13k_gemm_kq_w8pILi12ELb0EEv11gemm_params",
    "_Z12k_gemm_kq_w8ILi12ELb0ELi20EEv11gemm_params",
]


@pytest.mark.parametrize("kernel", SHIPPED)
def test_shipped_gemm_kernels_do_not_spill(gemm_asm, kernel):
    assert _prop(gemm_asm, kernel, "private_seg_size") == 0
    assert _lds(gemm_asm, kernel) <= 160 * 1024


def test_64x128_wave_tile_kernel_resources(tmp_path):
    """gemm_q_t64.hip: 8 waves = 2 per SIMD -> at most 256 registers; no scratch access inside the main loop (the 256-row form
    parks one float4 in scratch in its EPILOGUE, which costs nothing); ring + reduction area within 160 KB of LDS"""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    asm = _isa_tool().asm_of("gemm_q_t64.hip")

    def body_of(k):                                                     # (a 24-MB text: plain finds, not a DOTALL regex per kernel)
        i = asm.index("\n" + k + ":")
        return asm[i + 1:asm.index("\n.Lfunc_end", i)]
    # the plain product (TAIL = false; the tail-carrying twin shares the main loop) and — round 5 — the one-launch step that carries the activation quantizer and the
    # grid barrier in its prologue (FQ = true): the loop must be the same loop, in particular without a scratch access in it (a spill there would also break the
    # kernel's counted vmcnt waits, which assume that LDS-DMA is the only vector-memory traffic of the loop)
    # (type 102 = Q4_0R, round 5: Q4_0 through its resident 16-byte-aligned image — the same loop with a two-instruction constant step)
    for ty, tm, fq, max_scratch in ((12, 128, 0, 0), (12, 256, 0, 128), (12, 128, 1, 0), (102, 128, 0, 0), (102, 256, 0, 128)):
        k = "_Z13k_gemm_kq_t64ILi%dELi%dELb0ELi0ELb0ELb%dEEv11gemm_params" % (ty, tm, fq)
        assert _prop(asm, k, "num_vgpr") + _prop(asm, k, "num_agpr") <= 256
        assert _prop(asm, k, "private_seg_size") <= max_scratch
        assert _lds(asm, k) <= 160 * 1024
        body = body_of(k)
        # the steady-state stage pairs — one copy of the loop for the four loader waves, one for the others: no scratch access inside
        # (the 256-row form parks a few loader-only address registers in scratch AROUND the loops), 32 / 64 MFMAs each
        # (a loop = from its header to the first backward branch BEFORE the next loop's header: the quantizer's rolled loop in the FQ prologue ends in another branch form)
        loops = _loops(body)
        main = [lp for lp in loops if lp.count("v_mfma_f32_32x32x16_f16") == (32 if tm == 128 else 64)]
        assert len(main) == 2 and all("scratch_" not in lp for lp in main)
        assert sorted(lp.count("global_load_lds_dwordx4") for lp in main)[0] == 0        # the non-loader copy issues no LDS-DMA at all
    # the grouped MUL_MAT_ID instantiations (IDS; Q4_K and — round 5 — Q4_0R on a resident image of the expert stack): no scratch access in any loop that issues MFMAs
    for ty in (12, 102):
        for tm in (128, 256):
            k = "_Z13k_gemm_kq_t64ILi%dELi%dELb1ELi0ELb0ELb0EEv11gemm_params" % (ty, tm)
            assert _prop(asm, k, "num_vgpr") + _prop(asm, k, "num_agpr") <= 256 and _lds(asm, k) <= 160 * 1024
            body = body_of(k)
            loops = _loops(body)
            mf = [lp for lp in loops if "v_mfma_f32_32x32x16_f16" in lp]
            assert len(mf) >= 2 and all("scratch_" not in lp for lp in mf), k
    # nothing in this file loads into registers asynchronously: round 2's first version did (superblock headers, inline-asm
    # global_load_dwordx4 waited for a stage later) and hipcc copied the in-flight registers before the wait — one wave in a few
    # thousand got garbage constants on the GPU, invisibly to the CPU emulator.  Headers go through LDS (DMA) now.
    # (round 5: the one-launch step's prologue loads the fp32 activations into registers — compiler-issued, compiler-counted loads in front of the grid barrier, never
    #  beside the loop's hand-counted LDS-DMA: every one of them precedes the kernel's first MFMA; the kernels without the quantizer still have none)
    for name in sorted(set(re.findall(r"^(_Z13k_gemm_kq_t64\w+):", asm, re.M))):
        kbody = body_of(name)
        regloads = [m.start() for m in re.finditer(r"global_load_dwordx[24] v", kbody)]
        if name.endswith("ELb1EEv11gemm_params"):
            assert regloads and max(regloads) < kbody.index("v_mfma_f32_32x32x16_f16"), name
        else:
            assert not regloads, name


@pytest.mark.parametrize("m,k,b,splitk,tm", [(128, 256, 128, 1, 128), (300, 1536, 200, 1, 128), (300, 1536, 200, 1, 256), (513, 1024, 129, 2, 128),
                                             (256, 1792, 128, 2, 128), (512, 1792, 200, 2, 256), (200, 256, 100, 1, 256),
                                             (200, 2304, 40, 4, 128), (128, 4352, 9, 8, 128), (300, 2048, 130, 8, 128)])      # deep split: 9 / 17 / 8 superblocks over 4 / 8 / 8 work-groups
def test_64x128_wave_tile_kernel_source_on_the_cpu(m, k, b, splitk, tm):
    """tools/emul/t64_emul: the source of k_gemm_kq_t64 executed on the CPU against a direct fp16 product — both tile heights,
    ragged edges, one-superblock K ranges, even and uneven hand-off splits (the harness also checks that every exchange flag
    was reset by its reader) — with the LDS-DMA landing immediately and as late as the counted waits allow"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import importlib.util
    spec = importlib.util.spec_from_file_location("emul_check", os.path.join(ROOT, "tools", "emul", "emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for defer in (False, True):
        assert mod.run(m, k, b, seed=m + k, timeout=900, splitk=splitk, kernel="t64", exp=tm, defer_dma=defer) < 1e-6


@pytest.mark.parametrize("m,k,b,e", [(128, 512, 384, 2), (256, 1024, 512, 3)])
def test_grouped_mul_mat_id_kernel_source_on_the_cpu(m, k, b, e):
    """k_gemm_kq_t64<Q4_K, 128, IDS> (the grouped MUL_MAT_ID launch): per-tile expert matrices, an unused tile whose work-groups exit, padding
    rows that are not stored, output rows scattered through row_dst — against the per-expert fp16 product, DMA landing immediately and late"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import importlib.util
    spec = importlib.util.spec_from_file_location("emul_check", os.path.join(ROOT, "tools", "emul", "emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for defer in (False, True):
        err, untouched = mod.run_ids(m, k, b, e, seed=m + e, defer_dma=defer)
        assert err < 1e-6 and untouched


def test_64x128_wave_tile_kernel_counted_waits_are_tight(monkeypatch):
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import importlib.util
    spec = importlib.util.spec_from_file_location("emul_check", os.path.join(ROOT, "tools", "emul", "emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setenv("EMU_WEAKEN_WAITS", "1")
    assert mod.run(300, 1536, 200, seed=3, timeout=900, splitk=1, kernel="t64", exp=128, defer_dma=True) > 1e-3


def test_loader_wave_kernel_fits_three_waves_per_simd(gemm_asm):
    # 12 waves per work-group = 3 per SIMD: 512 registers / 3, allocation granule 8
    for k in ("_Z13k_gemm_kq_w12ILi12ELb1ELi0EEv11gemm_params", "_Z13k_gemm_kq_w12ILi202ELb1ELi0EEv11gemm_params",
              "_Z13k_gemm_kq_w12ILi208ELb1ELi0EEv11gemm_params", "_Z13k_gemm_kq_w12ILi214ELb1ELi0EEv11gemm_params"):
        assert _prop(gemm_asm, k, "num_vgpr") + _prop(gemm_asm, k, "num_agpr") <= 168


def test_two_waves_per_simd_kernels_fit_256_registers(gemm_asm):
    for k in SHIPPED[1:]:
        assert _prop(gemm_asm, k, "num_vgpr") + _prop(gemm_asm, k, "num_agpr") <= 256


@pytest.mark.parametrize("m,k,b,splitk,exp,l2", [(300, 1536, 200, 1, 0, 1), (256, 2048, 128, 2, 0, 1), (256, 2048, 128, 2, 0, 0), (256, 1024, 128, 1, 100, 1),
                                              (256, 1792, 128, 2, 0, 1), (300, 2304, 200, 2, 0, 0),          # ODD superblock counts: uneven 4 / 3 and 5 / 4 hand-off splits (the default route for odd counts)
                                              (513, 3072, 129, 2, 1, 1), (513, 3072, 129, 2, 2, 0), (300, 1536, 200, 1, 4, 1), (256, 2048, 128, 2, 6, 1)])
def test_shipped_12_wave_kernel_source_and_its_candidates_on_the_cpu(m, k, b, splitk, exp, l2):
    """tools/emul/w12_emul: the source of k_gemm_kq_w12 (+ the shared epilogue) executed on the CPU.  exp 0 is the shipped
    kernel (GPU-verified: this checks the emulator against it), 100 the variant without the loaders' scale table, 1 / 2 / 4 / 6
    the candidates kept in ablation builds (early table read, balanced epilogue, stores from registers, both): each must give
    the shipped kernel's result BIT FOR BIT, through both exchange transports"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("emul_check", os.path.join(ROOT, "tools", "emul", "emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    err, y = mod.run(m, k, b, seed=5, timeout=600, splitk=splitk, kernel="w12", exp=exp, xchg_l2=l2, return_y=True)
    assert err < 1e-6
    if exp != 0:
        _, y0 = mod.run(m, k, b, seed=5, timeout=600, splitk=splitk, kernel="w12", exp=0, xchg_l2=1, return_y=True)
        assert np.array_equal(y, y0)


@pytest.mark.parametrize("kernel,m,k,b,splitk,exp", [("w12", 300, 1536, 200, 1, 0), ("w12", 256, 2048, 128, 2, 0)])
def test_counted_vmcnt_waits_are_sufficient_and_tight(kernel, m, k, b, splitk, exp, monkeypatch):
    """EMU_DEFER_DMA=1: every LDS-DMA copy lands as LATE as the hardware permits — only when an s_waitcnt vmcnt(n) of the issuing
    wave retires it, in order — so a missing or too-weak wait leaves stale bytes in LDS.  Both kernels pass as written, and fail
    as soon as every wait tolerates one more outstanding operation (the self-test of this check)"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import importlib.util
    spec = importlib.util.spec_from_file_location("emul_check", os.path.join(ROOT, "tools", "emul", "emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(m, k, b, seed=3, timeout=600, splitk=splitk, kernel=kernel, exp=exp, defer_dma=True) < 1e-6
    monkeypatch.setenv("EMU_WEAKEN_WAITS", "1")
    assert mod.run(m, k, b, seed=3, timeout=600, splitk=splitk, kernel=kernel, exp=exp, defer_dma=True) > 1e-3


def test_no_out_of_bounds_access_at_the_shape_that_faulted_on_the_gpu(tmp_path):
    """DESIGN.md 4.3, open issue: a gemm_bench process at 8192 x 8192 x 512 died with a GPU fault.  The emulator places every
    global buffer between inaccessible pages; the first and the last work-group of the shipped kernel at that shape (64 stages
    each, full-size W / activation image / Y) run without touching them"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("emul_check", os.path.join(ROOT, "tools", "emul", "emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    M, K, B = 8192, 8192, 512
    rng = np.random.default_rng(0)
    nb = M * K // 256
    w = rng.integers(0, 256, (nb, 144), dtype=np.uint8)
    w[:, 0:2] = rng.uniform(0.001, 0.004, nb).astype(np.float16).view(np.uint8).reshape(nb, 2)
    w[:, 2:4] = rng.uniform(0.01, 0.03, nb).astype(np.float16).view(np.uint8).reshape(nb, 2)
    w.tofile(str(tmp_path / "w.bin"))
    rng.uniform(-1, 1, (K // 128, B, 128)).astype(np.float16).tofile(str(tmp_path / "xh.bin"))
    exe = mod.build("w12")
    for blocks in ("0:1", "255:256"):
        r = subprocess.run([exe, str(M), str(K), str(B), str(tmp_path / "w.bin"), str(tmp_path / "xh.bin"), str(tmp_path / "y.bin"), "1", "0", "1"],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, EMU_BLOCKS=blocks))
        if r.returncode == 77:
            pytest.skip("the environment cannot host the emulation (process / thread limits)")
        assert r.returncode == 0, (blocks, r.stderr[-300:])
    y = np.fromfile(str(tmp_path / "y.bin"), np.float32).reshape(B, M)
    assert np.isfinite(y[-128:, -128:]).all() and not (y[-128:, -128:] == -12345.0).any()      # the last work-group's tile was written


def _reads_before_wait(asm, kernel):
    """inline-asm register loads (global_load_dwordx4 into VGPRs) whose destination is read by ANY later instruction before
    the next s_waitcnt vmcnt — the compiler treats an asm output as available at once, the hardware does not interlock"""
    m = re.search(r"^%s:.*?^\.Lfunc_end" % re.escape(kernel), asm, re.S | re.M)
    assert m, kernel
    lines = [ln.split(";")[0].rstrip() for ln in m.group(0).split("\n") if ln.strip() and not ln.strip().startswith(";")]
    n, bad = 0, []
    for i, ln in enumerate(lines):
        mm = re.match(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\], v", ln)
        if not mm:
            continue
        n += 1
        regs = set(range(int(mm.group(1)), int(mm.group(2)) + 1))
        for t in lines[i + 1:i + 400]:
            if "s_waitcnt" in t and "vmcnt" in t:
                break
            parts = t.split(None, 1)
            if len(parts) < 2:
                continue
            ops = [x.strip() for x in parts[1].split(",")]
            srcs = ops if parts[0].startswith(("global_load_lds", "s_", "ds_write", "global_store", "buffer_store")) else ops[1:]
            used = set()
            for x in srcs:
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", x):
                    used |= set(range(int(a), int(b) + 1))
                used |= {int(a) for a in re.findall(r"\bv(\d+)\b", x)}
            if used & regs:
                bad.append((ln.strip(), t.strip()))
                break
    return n, bad


def test_asynchronous_register_loads_are_not_read_before_their_wait(gemm_asm):
    """k_gemm_kq_w12's loader waves load superblock headers into registers with inline asm and wait for them later with a
    vmcnt wait tied to those registers.  Nothing stops the compiler from copying the registers in between (it did, in a first
    cut of the experimental kernel); the shipped binary must be free of such reads"""
    ks = re.findall(r"^(_Z13k_gemm_kq_w12ILi\d+ELb1ELi0EEv11gemm_params):", gemm_asm, re.M)
    assert len(ks) >= 4, ks                                    # Q4_K and the three staged forms (Q4_0, Q8_0, Q6_K re-laid by the loader waves)
    for k in ks:
        n, bad = _reads_before_wait(gemm_asm, k)
        assert n > 0 and not bad, (k, bad[:3])


@pytest.mark.parametrize("ncol", [2, 3, 8])
@pytest.mark.parametrize("t", [12, 13, 14, 2, 8])
def test_small_batch_decode_kernel_source_on_the_cpu(t, ncol):
    """tools/emul/gemv_emul: k_gemv_q_fused<.., NB> for 2..8 activation rows in ONE launch (every work-group quantizes the rows into
    LDS — bit-exact quantizer bodies — and each weight unit meets all columns from there), the five main formats, against the oracle;
    3 rows run the 4-column instantiation with its padding column repeating the last row"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import importlib.util
    spec = importlib.util.spec_from_file_location("gemv_emul_check", os.path.join(ROOT, "tools", "emul", "gemv_emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(t, 37, 2048, seed=t + ncol, ncol=ncol) < 1e-5
    # ... and the form that is handed pre-quantized rows (one quantize launch for all work-groups) and only copies them into LDS
    assert mod.run_cols(t, 37, 2048, ncol, seed=t + ncol, staged=True) < 1e-5


@pytest.mark.parametrize("cfg", [0, 1])                       # 4 waves x 1 row, 8 waves x 2 rows (the M >= 4096 default)
@pytest.mark.parametrize("t", [12, 13, 14, 2, 8, 6, 10, 11, 3, 7, 20, 23])  # Q4_K, Q5_K, Q6_K, Q4_0, Q8_0, Q5_0, Q2_K, Q3_K, Q4_1, Q5_1 (in-launch Q8_1 quantizer), IQ4_NL, IQ4_XS
def test_decode_kernel_source_on_the_cpu(t, cfg):
    """tools/emul/gemv_emul: the source of the one-launch decode step (k_gemv_q_fused: in-kernel Q8_K / Q8_0 activation quantizer,
    int8 dots, wave reduction) executed on the CPU against the oracle's MUL_MAT — a GPU-free regression check of the B = 1 path"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import importlib.util
    spec = importlib.util.spec_from_file_location("gemv_emul_check", os.path.join(ROOT, "tools", "emul", "gemv_emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(t, 37, 2048, seed=t + cfg, env={"CDNA4_FUSED_CFG": str(cfg)}) < 1e-5


@pytest.mark.parametrize("k", [2048, 12288])                  # one round of 64 units / three rounds (the register sets trade places twice)
@pytest.mark.parametrize("t", [12, 13, 14, 2, 10, 11, 23])    # the formats whose launcher can choose 16 waves x 1 row (K-quants and Q4_0)
@pytest.mark.parametrize("cfg", [4, 1, 3])                    # 16 x 1 (round 5), 8 x 2 (swapped loop), 8 x 1 (copying loop)
def test_decode_kernel_configurations_on_the_cpu(t, cfg, k):
    """round 5's decode configurations: 1024-thread work-groups with one row per wave, and the two forms of the multi-round loop (alternating register sets / copies),
    on rows of one and of three rounds — the kernel's source against the oracle's MUL_MAT"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    if k > 2048 and t not in (12, 14, 2):
        pytest.skip("the long rows: three formats are enough (emulation time)")
    import importlib.util
    spec = importlib.util.spec_from_file_location("gemv_emul_check", os.path.join(ROOT, "tools", "emul", "gemv_emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(t, 19, k, seed=t + cfg, env={"CDNA4_FUSED_CFG": str(cfg)}) < 1e-5


@pytest.mark.parametrize("kern,wtype,m,k,b,splitk", [(20, 12, 300, 1536, 200, 1), (20, 12, 256, 2048, 128, 2), (64, 12, 300, 1536, 200, 1), (64, 13, 256, 2048, 128, 2),
                                                     (64, 13, 300, 1536, 200, 1), (20, 13, 300, 512, 200, 1), (1064, 12, 256, 2048, 128, 2), (64, 13, 256, 1792, 128, 2)])
def test_8_wave_kernel_sources_on_the_cpu(kern, wtype, m, k, b, splitk):
    """tools/emul/w8_emul: k_gemm_kq_w8 (schedule 20: the shallow-K fallback and the kernel of the repacked formats) and
    k_gemm_kq_w8p (64: Q5_K's default; 1064: its TRACE build), Q4_K and Q5_K weights, executed on the CPU — with the LDS-DMA
    landing immediately and as late as the counted waits allow"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import importlib.util
    spec = importlib.util.spec_from_file_location("emul_check", os.path.join(ROOT, "tools", "emul", "emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for defer in (False, True):
        assert mod.run(m, k, b, seed=m + k, timeout=600, splitk=splitk, kernel="w8", exp=kern, wtype=wtype, defer_dma=defer) < 1e-6


@pytest.mark.parametrize("kind,k,b,dist", [(0, 4096, 8, "uniform"), (0, 2048, 32, "ties"), (1, 4096, 8, "uniform"), (1, 1024, 32, "ties"), (2, 1024, 32, "ties"),
                                           (3, 2048, 8, "uniform"), (3, 1024, 16, "ties")])       # 3 = Q8_1 (k_quantize_q8_1: d, s = fp16(d * sum q), quants, image)
def test_activation_quantizer_sources_on_the_cpu_bit_exact(kind, k, b, dist):
    """tools/emul/quant_emul: k_quantize_q8_K / k_quantize_q8_0 (AVX2 and _ref roundings) executed on the CPU equal the oracle's
    quantize_row_q8_K / _q8_0 bit for bit — quants, scales, bsums — and their fp16 activation image equals fp16(d*q) in the
    panel-major, pair-interleaved layout; all-zero blocks, a negative maximum and exact .5 ties included"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import importlib.util
    spec = importlib.util.spec_from_file_location("quant_emul_check", os.path.join(ROOT, "tools", "emul", "quant_emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(kind, k, b, dist=dist, seed=kind + k)


@pytest.mark.parametrize("t", [12, 13, 14, 2, 8, 6, 10, 11, 3, 7, 20, 23])
def test_multi_column_gemv_source_on_the_cpu(t):
    """tools/emul/gemv_emul: k_gemv_q (2 <= B <= 8 columns share one pass over the weights) on activations quantized by the
    oracle, against the oracle's MUL_MAT"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import importlib.util
    spec = importlib.util.spec_from_file_location("gemv_emul_check", os.path.join(ROOT, "tools", "emul", "gemv_emul_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run_cols(t, 33, 2048, 3, seed=t) < 1e-5
    assert mod.run_cols(t, 16, 1024, 8, seed=t + 1) < 1e-5


def _emul_module(name):
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm clang not available")
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "emul", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("wtype", [2, 8, 14])         # ggml type ids: Q4_0, Q8_0, Q6_K
def test_r8_on_the_resident_relayouts_counted_waits_are_sufficient_and_tight(wtype):
    """k_gemm_r8<Q4_0R | Q8_0R | Q6_K8> (round 5) on the CPU with every LDS-DMA copy performed as LATE as its counted vmcnt wait allows: correct (Q8_0R issues TWO raw pieces per K
    tile, so the loop's wait tolerates two outstanding operations instead of one); with every wait weakened by one the result is wrong — the waits are not slack.  Also the
    reduce-scatter split in two and a ragged tile."""
    mod = _emul_module("emul_check")
    assert mod.run_relayout(300, 1280, 200, wtype, defer_dma=True) < 1e-6
    assert mod.run_relayout(256, 1024, 256, wtype, splitk=2, defer_dma=True) < 1e-6
    assert mod.run_relayout(256, 768, 256, wtype, defer_dma=True, weaken=1) > 1e-2


@pytest.mark.parametrize("t,m,k", [(6, 8, 1024), (6, 33, 64), (11, 8, 1024), (11, 33, 512), (10, 8, 1024), (10, 33, 512), (10, 5, 256),
                                   (20, 8, 1024), (20, 33, 64), (3, 8, 1024), (3, 33, 128), (7, 8, 1024), (7, 33, 128), (23, 8, 1024), (23, 33, 256)])
def test_weight_reencoding_sources_on_the_cpu_are_exact(t, m, k):
    """tools/emul/convert_emul: k_convert_q5_0_q8_0 / k_convert_q3_K_q6_K / k_convert_q2_K_q6_K2 (the prefill route of Q5_0 / Q3_K / Q2_K)
    executed on the CPU: the oracle's dequantize_row of the re-encoded matrix equals its dequantize_row of the source bit for bit, on fully
    random block bytes; for Q2_K the sum of its two parts (scale part + minimum part, 2 K columns) equals it value for value.  Likewise IQ4_NL -> Q8_0
    (k_convert_iq4_nl_q8_0), Q4_1 / Q5_1 -> [d q | m 1] in Q8_0 (k_convert_q41_q8_0x2: the sum of the two parts bit for bit) and IQ4_XS -> [h part | l part]
    in Q6_K (k_convert_iq4_xs_q6_K2: value for value)"""
    assert _emul_module("convert_emul_check").run(t, m, k, seed=t + k)


@pytest.mark.parametrize("t", [2, 3, 6, 7, 8])
def test_fp16_copy_of_a_quantized_kv_source_on_the_cpu_bit_exact(t):
    """tools/emul/deq_emul f16: k_q_to_f16_dense (the pass in front of FLASH_ATTN_EXT when K / V arrive quantized) on strided rows equals
    fp16(oracle to_float) bit for bit, densely packed"""
    assert _emul_module("deq_emul_check").run_f16(t, 128, seed=t) and _emul_module("deq_emul_check").run_f16(t, 64, nrows=4, gap=0, seed=t + 1)


@pytest.mark.parametrize("t", [2, 8, 3, 6, 7])
def test_cpy_quantizer_sources_on_the_cpu_byte_exact(t):
    """tools/emul/deq_emul cpyq: ggml_cdna4_op_cpy F32 -> Q4_0 / Q8_0 / Q4_1 / Q5_0 / Q5_1 (k_cpy_f32_to_q, k_cpy_f32_to_q45) executed on the CPU equals the
    oracle's quantize_row_*_ref byte for byte (all-zero, constant and negative-maximum blocks included)"""
    mod = _emul_module("deq_emul_check")
    assert mod.run_cpyq(t, seed=t) and mod.run_cpyq(t, 512, 7, "normal", seed=t + 1)


@pytest.mark.parametrize("t", [2, 3, 6, 7, 8, 10, 11, 12, 13, 14, 20, 23])
def test_to_float_sources_on_the_cpu_bit_exact(t):
    """tools/emul/deq_emul: deq_elem of ops.hip (dequantize_row, GET_ROWS, CPY -> F32) executed on the CPU equals the oracle's dequantize_row_*
    bit for bit for all twelve block formats, on fully random block bytes"""
    assert _emul_module("deq_emul_check").run(t, 32 * 256 if t > 9 else 32 * 24, seed=t)


FATTN_EMUL_CASES = [dict(D=64, n_q=5, n_head=2, n_kv=96), dict(D=128, n_q=35, n_head=4, n_kv=200, n_head_kv=2, max_bias=8.0), dict(D=128, n_q=35, n_head=4, n_kv=200, n_head_kv=2, max_bias=8.0, cus=2),
                    dict(D=128, n_q=3, n_head=2, n_kv=64, softcap=10.0), dict(D=256, n_q=33, n_head=2, n_kv=130, n_head_kv=1, mask=False, cus=1),
                    dict(D=64, n_q=1, n_head=3, n_kv=517, inf_every=7), dict(D=128, n_q=40, n_head=2, n_kv=300, n_batch=2, permuted=True, cus=4),
                    dict(D=64, n_q=32, n_head=2, n_kv=31), dict(D=256, n_q=1, n_head=2, n_kv=1024, max_bias=8.0, inf_every=5), dict(D=128, n_q=2, n_head=2, n_kv=2048, inf_every=3),
                    dict(D=64, n_q=130, n_head=2, n_kv=257, inf_every=4), dict(D=64, n_q=130, n_head=2, n_kv=256, inf_every=4, cus=2),
                    dict(D=64, n_q=70, n_head=2, n_kv=1024, inf_every=4, max_bias=8.0), dict(D=256, n_q=200, n_head=1, n_kv=96, n_batch=2, cus=2),
                    dict(D=128, n_q=512, n_head=2, n_kv=512, n_head_kv=1, cus=8, inf_every=3)]


@pytest.mark.parametrize("kw", FATTN_EMUL_CASES)
def test_flash_attn_source_on_the_cpu(kw):
    """tools/emul/fattn_emul: k_flash_attn_split (+ k_flash_attn_merge when the keys are split over work-groups: the n_kv >= 512 cases; one or
    several 32-row query tiles) and k_flash_attn_wide (the `cus` cases — chosen when 128-row tiles fill the chip: K / V chunks and the
    transposed V fragments shared through LDS); masks on the vector path (16-byte aligned rows, transposed by the matrix core, -inf
    included) and on the element path (n_kv not a multiple of 8, ragged last chunk) — S^T = K.Q^T, V transposed on the matrix
    core, O^T += Vt^T.P^T — executed on the CPU with the MFMA emulated lane for lane: <= 5e-4 from a float64 evaluation of the operator, <= 6e-3 from the
    oracle (whose FP16 accumulator — the reference's, ggml-cpu.c:10960-10974 — is the larger part of that distance).  Grouped-query heads,
    ALiBi, softcap, no mask, -inf mask entries incl. a fully masked stretch, batches, permuted operands, ragged n_q / n_kv."""
    r = _emul_module("fattn_emul_check").run(**kw)
    if r is None:
        pytest.skip("the environment cannot host the emulation")
    assert r[0] < 5e-4 and r[1] < 6e-3, r


@pytest.mark.parametrize("t,kw", [(8, dict(D=64, n_q=5, n_head=2, n_kv=96)), (2, dict(D=128, n_q=3, n_head=4, n_kv=70, n_head_kv=2, permuted=True)),
                                  (3, dict(D=256, n_q=40, n_head=2, n_kv=200, cus=1)), (6, dict(D=128, n_q=1, n_head=4, n_kv=1024, n_head_kv=1)), (7, dict(D=64, n_q=33, n_head=2, n_kv=130))])
def test_flash_attn_on_a_quantized_kv_end_to_end_on_the_cpu(t, kw):
    """tools/emul/fattn_emul with a block-quantized K / V: ggml_cdna4_op_flash_attn_ext's own host code (scratch, descriptors), the conversion pass
    k_q_to_f16_dense and the F16 kernels, all from source on the CPU — <= 5e-4 from a float64 evaluation on the dequantized K / V and bit-identical
    to the F16 path on a cache that holds fp16(to_float(K)), fp16(to_float(V)); strided (permuted) rows, grouped-query heads, both kernels"""
    r = _emul_module("fattn_emul_check").run_quantized(t, seed=t, **kw)
    if r is None:
        pytest.skip("the environment cannot host the emulation")
    assert r[0] < 5e-4 and r[1] is True, r


@pytest.mark.parametrize("kw", [dict(D=80, n_q=5, n_head=2, n_kv=96), dict(D=80, n_q=35, n_head=4, n_kv=200, n_head_kv=2, max_bias=8.0, cus=2), dict(D=80, n_q=3, n_head=2, n_kv=70, permuted=True, n_batch=2),
                                dict(D=96, n_q=33, n_head=2, n_kv=130, mask=False), dict(D=112, n_q=1, n_head=3, n_kv=517, inf_every=7), dict(D=40, n_q=4, n_head=2, n_kv=64), dict(D=200, n_q=7, n_head=2, n_kv=90)])
def test_flash_attn_padded_head_sizes_end_to_end_on_the_cpu(kw):
    """head sizes without a kernel of their own (80 — the stock harness's —, 96, 112, 40, 200) run zero-padded to 64 / 128 / 256: the op's host code
    (padded copies of q / k / v, the result copied back) and the kernels from source on the CPU, same bars as the native head sizes"""
    r = _emul_module("fattn_emul_check").run(**kw)
    if r is None:
        pytest.skip("the environment cannot host the emulation")
    assert r[0] < 5e-4 and r[1] < 6e-3, r


def test_flash_attn_padded_head_size_with_a_quantized_kv_on_the_cpu():
    mod = _emul_module("fattn_emul_check")
    for t, D in ((8, 96), (2, 96), (7, 160)):
        r = mod.run_quantized(t, D, 5, 2, 96, seed=t)
        if r is None:
            pytest.skip("the environment cannot host the emulation")
        assert r[0] < 5e-4 and r[1] is True, (t, D, r)


def test_hardware_verified_kernels_are_unchanged():
    """tools/isa_manifest.py: the ISA of every kernel that was part of a build with a green hardware session (profiles/rNN/isa_manifest.json,
    `hw`: true) is what hipcc emits for it today — so what was added since (listed there as `hw`: false, or new) cannot have changed the
    kernels the MI355X numbers and parity results belong to.  Comparable only under the compiler that wrote the record."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    import glob
    import importlib.util
    mans = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "isa_manifest.json")))
    assert mans, "no profiles/rNN/isa_manifest.json"
    spec = importlib.util.spec_from_file_location("isa_manifest", os.path.join(ROOT, "tools", "isa_manifest.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ok, changed_hw, changed, new, gone = mod.check(mans[-1])
    if not ok:
        pytest.skip("the manifest was written under another compiler version")
    gone_hw = [g for g in gone if json_hw(mans[-1], g)]
    if changed_hw or gone_hw:
        # Round 3 works on the kernels themselves with the GPU in the loop: a changed kernel is not an error, it is a kernel whose evidence is
        # the NEXT hardware session (scripts/gpu_tests_only.sh), after which the manifest is rewritten.  The emulator is a lint, not a gate.
        pytest.skip("%d hardware-verified kernels changed / %d gone since %s was written (e.g. %s): their evidence is the next GPU session"
                    % (len(changed_hw), len(gone_hw), os.path.relpath(mans[-1], ROOT), (changed_hw + gone_hw)[:3]))


def json_hw(path, fk):
    import json
    return json.load(open(path))["kernels"][fk[0]][fk[1]]["hw"]


# ---- the WHOLE library on the CPU (tools/emul/lib_emul.h): ggml_cdna4_mul_mat / _mul_mat_id through the C-ABI of a host build of the product's own sources.
# (Wave-collectives complete among the lanes that execute them — tools/emul/hip_emul.h, Sync — so any shape runs, partial waves included.)
LIB_TYPES = [("q4_0", 2), ("q4_1", 3), ("q5_0", 6), ("q5_1", 7), ("q8_0", 8), ("q2_K", 10), ("q3_K", 11), ("q4_K", 12), ("q5_K", 13), ("q6_K", 14), ("iq4_nl", 20), ("iq4_xs", 23)]


@pytest.mark.parametrize("name,t", LIB_TYPES)
def test_whole_library_mul_mat_on_the_cpu(name, t):
    """every accepted weight type through ggml_cdna4_mul_mat on the CPU, the three regimes of the AUTO route: one row (quantizer inside the GEMV
    launch), four rows (quantize + GEMV; Q8_1 activations for Q4_1 / Q5_1), 32 rows (the MFMA GEMM: its own kernel for the five headline formats —
    k_gemm_kq_t64 / w8p / the staging w12 with a split-K hand-off —, an exact re-encoding for the others: Q5_0 / IQ4_NL -> Q8_0, Q3_K -> Q6_K, and
    the two-part forms of Q2_K / Q4_1 / Q5_1 / IQ4_XS against the doubled activation image) — the host code between the C-ABI and the kernels
    included.  GEMV <= 1e-5, GEMM <= 1e-3 from the oracle's MUL_MAT of that type."""
    mod = _emul_module("lib_emul_check")
    for m, k, b, bar in ((40, 1024, 1, 1e-5), (24, 1024, 4, 1e-5), (130, 768, 32, 1e-3)):
        r = mod.mul_mat(t, m, k, b, seed=t + b, timeout=300)
        if r is None:
            pytest.skip("the environment cannot host the emulation")
        assert r[0] < bar, (name, m, k, b, r[0])


@pytest.mark.parametrize("t", [12, 13])
@pytest.mark.parametrize("m,k,b,cus", [(512, 1024, 256, 2), (300, 512, 300, 2), (256, 2048, 200, 1)])
def test_whole_library_large_grid_route_on_the_cpu(m, k, b, cus, t):
    """Q4_K / Q5_K where the 256 x 256 tiles of k_gemm_r8 fill (a pretend chip of `cus` CUs): AUTO takes that kernel, unsplit — two tiles on 2 CUs, four
    ragged tiles on 2 CUs, one tile on 1 CU (its split-K exchange: tools/emul/emul_check.py, kernel="lds") — through the C-ABI on the CPU, within the GEMM bar of the oracle"""
    r = _emul_module("lib_emul_check").mul_mat(t, m, k, b, seed=m + b, cus=cus, timeout=900)
    if r is None:
        pytest.skip("the environment cannot host the emulation")
    assert r[0] < 1e-3, r[0]


@pytest.mark.parametrize("m,k,b,cus", [(512, 768, 80, 8), (512, 512, 100, 8), (512, 256, 100, 4)])
def test_whole_library_one_launch_step_on_the_cpu(m, k, b, cus):
    """round 5's one-launch step (k_gemm_kq_t64<.., FQ>: activation quantizer -> two-level grid barrier -> multiply) through the C-ABI on the CPU, on pretend chips whose
    grids are resident and whose quantizer share is one pass: split in two with an uneven superblock count (hand-off inside the launch), split in two evenly on 8 work-groups,
    unsplit — each bit-identical to the two launches it replaces (CDNA4_NO_FUSEQ=1), with the LDS-DMA landing immediately and as late as the counted waits allow"""
    mod = _emul_module("lib_emul_check")
    import numpy as np
    got = {}
    for tag, envs in (("one", {}), ("two", {"CDNA4_NO_FUSEQ": "1"}), ("one_deferred", {"EMU_DEFER_DMA": "1"})):
        old = {k_: os.environ.get(k_) for k_ in ("CDNA4_NO_FUSEQ", "EMU_DEFER_DMA")}
        os.environ.update(envs)
        try:
            r = mod.mul_mat(12, m, k, b, path=2, seed=m + b, cus=cus, timeout=900)
        finally:
            for k_, v_ in old.items():
                if v_ is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = v_
        if r is None:
            pytest.skip("the environment cannot host the emulation")
        assert r[0] < 1e-3, (tag, r[0])
        got[tag] = r[1]
    assert np.array_equal(got["one"].view(np.uint32), got["two"].view(np.uint32))
    assert np.array_equal(got["one"].view(np.uint32), got["one_deferred"].view(np.uint32))
    # and it IS the one-launch route on that chip (asked of the emulated library's own routing; in a process of its own: the CU count is cached per process)
    import subprocess, sys
    code = ("import ctypes, sys; so = ctypes.CDLL(sys.argv[1]); f = so.ggml_cdna4_mul_mat_route; f.argtypes = [ctypes.c_int] + [ctypes.c_int64] * 3; "
            "print(f(12, %d, %d, %d))" % (m, k, b))
    r = subprocess.run([sys.executable, "-c", code, mod.build_so()], capture_output=True, text=True, timeout=300, env=dict(os.environ, EMU_CUS=str(cus)))
    assert r.returncode == 0 and r.stdout.strip() == "11", (r.stdout, r.stderr[-500:])


@pytest.mark.parametrize("name,t", [("q4_1", 3), ("iq4_nl", 20), ("iq4_xs", 23), ("q2_K", 10)])
def test_whole_library_small_k_gemm_route_on_the_cpu(name, t):
    """K = 256: the re-encoded matrix is too shallow for the staging kernel (2 superblocks of the target format) and takes the re-layout + 8-wave
    kernel; the shape of the stock harness's MUL_MAT cases (m = 16, k = 256, n = 16 -> here 32 activation rows for whole waves)"""
    r = _emul_module("lib_emul_check").mul_mat(t, 16, 256, 32, seed=t, timeout=300)
    if r is None:
        pytest.skip("the environment cannot host the emulation")
    assert r[0] < 1e-3, r[0]


@pytest.mark.parametrize("name,t", [("q4_1", 3), ("q5_1", 7), ("iq4_nl", 20), ("iq4_xs", 23), ("q4_K", 12)])
def test_whole_library_mul_mat_id_on_the_cpu(name, t):
    """ggml_cdna4_mul_mat_id on the CPU: one token (one launch, quantizer inside, ids read on the device) and four tokens with a broadcast
    activation row (quantize + GEMV with per-column experts)"""
    mod = _emul_module("lib_emul_check")
    for n_b, n_tok in ((2, 1), (1, 4)):
        r = mod.mul_mat_id(t, 64, 1024, 4, 2, n_b, n_tok, seed=t + n_tok, timeout=300)
        if r is None:
            pytest.skip("the environment cannot host the emulation")
        assert r < 1e-5, (name, n_b, n_tok, r)


@pytest.mark.parametrize("t,m,k,ne,nu,nb,nt", [(12, 128, 512, 4, 2, 2, 32), (12, 40, 256, 8, 2, 1, 40), (13, 130, 512, 3, 2, 2, 40), (14, 64, 256, 4, 2, 1, 40), (2, 50, 512, 4, 2, 2, 33), (8, 64, 256, 4, 2, 2, 33),
                                                (12, 64, 512, 2, 1, 1, 60)])
def test_whole_library_few_rows_per_expert_mul_mat_id_on_the_cpu(t, m, k, ne, nu, nb, nt):
    """MUL_MAT_ID with more than 32 (token, slot) rows but at most 32 per expert on average: the int8 matrix-core kernels over 32-row chunks of the expert-sorted
    image (k_mmq_*<2, 8> with the grouped argument block: expert from tile_expert, output rows through row_dst, chunks past a run exit; the last case gives one
    expert ~30 rows: two chunks of its tile, the second ragged) — the CPU's integer block dots: <= 1e-5 from the oracle, where the grouped fp16 GEMM sits at 3e-4"""
    os.environ["CDNA4_MMQ_IDS"] = "2"                                 # (Q4_K keeps its grouped fp16 GEMM in AUTO — faster on MI355X, capi.hip; forced here so that its kernel is covered too)
    try:
        r = _emul_module("lib_emul_check").mul_mat_id(t, m, k, ne, nu, nb, nt, seed=7, timeout=900)
    finally:
        os.environ.pop("CDNA4_MMQ_IDS", None)
    if r is None:
        pytest.skip("the environment cannot host the emulation")
    assert r < 1e-5, r


@pytest.mark.parametrize("m,k,ne,nu,nb,nt,cus", [(300, 512, 3, 2, 2, 70, 2), (256, 1024, 4, 2, 1, 90, 4)])
def test_whole_library_grouped_mul_mat_id_on_256_row_tiles_on_the_cpu(m, k, ne, nu, nb, nt, cus):
    """the grouped Q4_K launch on k_gemm_kq_t64<.., 256, IDS> (taken once the grouped grid offers 3/4 of a 256-row tile per CU of a pretend small chip): ragged m, padding rows,
    broadcast activation rows — within the GEMM bar of the oracle's MUL_MAT_ID"""
    r = _emul_module("lib_emul_check").mul_mat_id(12, m, k, ne, nu, nb, nt, seed=3, cus=cus, timeout=900)
    if r is None:
        pytest.skip("the environment cannot host the emulation")
    assert r < 1e-3, r


@pytest.mark.parametrize("m,k,ne,nu,nb,nt", [(130, 1024, 3, 2, 2, 70), (256, 2048, 4, 2, 1, 40)])
def test_whole_library_grouped_mul_mat_id_with_the_ticketed_k_split_on_the_cpu(m, k, ne, nu, nb, nt):
    """the grouped Q4_K launch where its tiles outnumber (a pretend chip of 2) CUs: every tile is computed by TWO work-groups over half of K each; the
    last one to arrive adds both partial tiles in the order ks = 0, 1 and stores (nobody waits: the tiles need not be co-resident); ticket counters
    back at zero afterwards (the next call of the same process — the second size — starts from them)"""
    mod = _emul_module("lib_emul_check")
    os.environ["CDNA4_MOE_SPLITK"] = "2"                             # (opt-in: a measured loss on MI355X, gemm_q_t64.hip)
    try:
        r = mod.mul_mat_id(12, m, k, ne, nu, nb, nt, seed=5, cus=2, timeout=900)
    finally:
        os.environ.pop("CDNA4_MOE_SPLITK", None)
    if r is None:
        pytest.skip("the environment cannot host the emulation")
    assert r < 1e-3, r


@pytest.mark.parametrize("name,t", LIB_TYPES)
def test_whole_library_stock_harness_shapes_on_the_cpu(name, t):
    """the shapes of the reference's test-backend-ops MUL_MAT sweep (m = 16, k = 256, n = 1 / 9 / 16: tests/test-backend-ops.cpp:4005-4081) through the
    C-ABI on the CPU — quantizers with partly filled waves, the shallow-K GEMM routes, the two-part forms at their smallest size"""
    mod = _emul_module("lib_emul_check")
    for b in (1, 9, 16):
        r = mod.mul_mat(t, 16, 256, b, seed=t + b, timeout=300)
        if r is None:
            pytest.skip("the environment cannot host the emulation")
        assert r[0] < (1e-5 if b <= 8 else 1e-3), (name, b, r[0])


@pytest.mark.parametrize("name,t", [("q4_0", 2), ("q8_0", 8), ("q5_0", 6), ("q4_1", 3), ("q5_1", 7), ("iq4_nl", 20)])
@pytest.mark.parametrize("k", [544, 992, 96])
def test_32_weight_formats_with_k_not_a_multiple_of_64_on_the_cpu(name, t, k):
    """a bug of round 2's decode kernel that only the whole-library emulation showed: lds_swz() permutes 16-byte chunks of the int8 activation row
    inside groups of four, so with K % 64 == 32 and K mod 1024 >= 512 the row's last two chunks landed on its scales (K = 544: rel-L2 1e27).  Such
    K now take the quantize + GEMV pair (cdna4_gemv_fused_supported); one row, and MUL_MAT_ID's single-token form, through the C-ABI"""
    mod = _emul_module("lib_emul_check")
    r = mod.mul_mat(t, 16, k, 1, seed=k, timeout=300)
    if r is None:
        pytest.skip("the environment cannot host the emulation")
    assert r[0] < 1e-5, (name, k, r[0])
    assert mod.mul_mat_id(t, 32, k, 4, 2, 2, 1, seed=k + 1, timeout=300) < 1e-5


@pytest.mark.parametrize("t,m,k,ne,nu,nb,nt", [(12, 128, 512, 4, 2, 2, 32), (12, 64, 256, 8, 2, 1, 40), (12, 130, 768, 3, 2, 2, 70),
                                                (13, 130, 512, 3, 2, 2, 40), (14, 64, 256, 4, 2, 1, 40), (2, 130, 384, 3, 2, 2, 40), (8, 64, 256, 4, 2, 2, 33)])
def test_whole_library_grouped_mul_mat_id_on_the_cpu(t, m, k, ne, nu, nb, nt):
    """prefill-sized MUL_MAT_ID through the C-ABI on the CPU: the device-side counting sort of the expert ids (k_moe_plan), the gathering
    activation quantizer (Q8_K / Q8_0) and the grouped launch — k_gemm_kq_t64<.., IDS> for Q4_K, k_gemm_q<.., IDS> for Q5_K / Q6_K / Q4_0 / Q8_0 —
    ragged per-expert counts, padding rows, broadcast activation rows"""
    r = _emul_module("lib_emul_check").mul_mat_id(t, m, k, ne, nu, nb, nt, seed=5, timeout=300)
    if r is None:
        pytest.skip("the environment cannot host the emulation")
    assert r < 1e-3, r
