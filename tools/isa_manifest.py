"""Per-kernel fingerprints of the gfx950 ISA hipcc emits for the kernel sources, so that "this change did not touch a hardware-verified
kernel" is a checkable statement in sessions without a GPU.

    python tools/isa_manifest.py --write profiles/rNN/isa_manifest.json     # after a hardware session: record what was verified
    python tools/isa_manifest.py --check profiles/rNN/isa_manifest.json     # later: which recorded kernels changed, which are new

A fingerprint is the sha256 of a kernel's assembly body (comments, label numbers and spacing removed), compiled with the flags of
ggml_amd/build.py.  `hw` says whether the kernel had run on an MI355X (parity-green) when the manifest was written; tests/test_build_static.py
asserts that no `hw` kernel differs from its record.  The record is only comparable under the same compiler (its version is stored)."""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ggml_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FILES = ["quantize_act.hip", "gemv_q.hip", "mmq_i8.hip", "gemm_q_mfma.hip", "gemm_q_t64.hip", "gemm_q_sk.hip", "gemm_q_lds.hip", "exact.hip", "convert_w.hip", "ops.hip", "fattn.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-inline-asm", "-S", "--cuda-device-only"]
# kernels written after the LAST full hardware session of the round being recorded (emulator-verified only).  Round 2 ended with such a list (Q4_1 /
# Q5_1 / IQ4_* units, k_quantize_q8_1, the two-part re-encodings, k_q_to_f16_dense, k_cpy_f32_to_q45: see profiles/r02/isa_manifest.json); round 3's
# manifest was written from the build of its final run on HEAD (profiles/r03/pytest_gpu_final_results.txt: 740 passed, 8 skipped) — nothing is pending; round 4's likewise
# (profiles/r04/pytest_gpu_final.log)
# round 5: written after the round's GPU time ran out (emulator-verified, their GPU test is in the suite): the grouped MUL_MAT_ID kernel on a resident Q4_0R expert stack;
# never launched by a test on the GPU: k_gemm_r8's tail-carrying twins for the three resident re-layouts (their plain twins ran; the Q4_K / Q5_K tail twins ran)
# k_norm<.., 2> (the NORM chain that also leaves the Q8_0 activation image): emulator-verified; k_norm<.., 0 | 1> are the hardware-verified kernels under a new template
# signature (their ISA is identical to the verified build's apart from the mangled names of their LDS arrays — checked by compiling the previous commit's source)
# round 6 (profiles/r06/pytest_gpu_final.log: the full suite on the build this manifest records): k_norm<.., 2> ran; the per-tile grouped kernel on a Q4_0R expert stack is
# now BEHIND the work-queue form (k_gemm_kq_sk<Q4_0R> ran instead: tests/test_gpu_resident.py) and is reached only with CDNA4_MOE_SK=0 or by jobs the queue does not take —
# emulator-verified, not launched on the GPU; k_gemm_r8's tail-carrying twins for the three resident re-layouts: as in round 5
NOT_ON_HARDWARE_YET = [r"k_gemm_kq_t64ILi102ELi(128|256)ELb1", r"k_gemm_r8ILi(102|108|115)ELi0ELb1"]


def compiler_version():
    return subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.strip().splitlines()[0:2]


def asm_of(f):
    """the gfx950 assembly of one kernel source under FLAGS.  Cached under build/isa_asm/ keyed by the contents of every file of csrc/ (one compile of gemm_q_t64.hip takes
    ~110 s: the manifest check and the resource tests of tests/test_build_static.py read the same text)"""
    h = hashlib.sha256(" ".join(FLAGS + compiler_version()).encode())
    for name in sorted(os.listdir(CSRC)):
        q = os.path.join(CSRC, name)
        if os.path.isfile(q):
            h.update(name.encode()); h.update(open(q, "rb").read())
    cache = os.path.join(ROOT, "build", "isa_asm")
    os.makedirs(cache, exist_ok=True)
    out = os.path.join(cache, "%s.%s.s" % (f, h.hexdigest()[:16]))
    if not os.path.exists(out):
        for old in os.listdir(cache):
            if old.startswith(f + ".") and old.endswith(".s"):           # (never another process's output in flight, "<out>.tmp<pid>": xdist workers compile side by side)
                try: os.remove(os.path.join(cache, old))
                except FileNotFoundError: pass
        tmp = out + ".tmp%d" % os.getpid()
        subprocess.run([HIPCC] + FLAGS + ["-o", tmp, os.path.join(CSRC, f)], check=True, capture_output=True, timeout=1500)
        os.replace(tmp, out)
    return open(out).read()


def fingerprints(files=FILES, jobs=4):
    def one(f):
        txt = asm_of(f)
        res = {}
        for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
            body = re.sub(r";.*", "", m.group(2))
            body = re.sub(r"\.LBB\d+_\d+", "L", body)
            body = re.sub(r"\.Ltmp\d+", "T", body)
            body = re.sub(r"[ \t]+", " ", body)
            res[m.group(1)] = hashlib.sha256(body.encode()).hexdigest()[:24]
        return f, res
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        return dict(ex.map(one, files))


def write(path):
    fp = fingerprints()
    man = {"compiler": compiler_version(), "flags": FLAGS, "kernels": {}}
    for f, ks in fp.items():
        man["kernels"][f] = {k: {"sha": v, "hw": not any(re.search(p, k) for p in NOT_ON_HARDWARE_YET)} for k, v in sorted(ks.items())}
    json.dump(man, open(path, "w"), indent=0, sort_keys=True)
    n = sum(len(v) for v in man["kernels"].values()); nhw = sum(e["hw"] for v in man["kernels"].values() for e in v.values())
    print("wrote %s: %d kernels, %d marked hardware-verified" % (path, n, nhw))


def check(path, files=None):
    """-> (comparable, changed hw kernels, changed other kernels, new kernels, gone kernels)"""
    man = json.load(open(path))
    if man["compiler"] != compiler_version():
        return False, [], [], [], []
    fp = fingerprints(files or [f for f in FILES if f in man["kernels"]])
    changed_hw, changed, new, gone = [], [], [], []
    for f, ks in fp.items():
        rec = man["kernels"].get(f, {})
        for k, sha in ks.items():
            if k not in rec:
                new.append((f, k))
            elif rec[k]["sha"] != sha:
                (changed_hw if rec[k]["hw"] else changed).append((f, k))
        gone += [(f, k) for k in rec if k not in ks]
    return True, changed_hw, changed, new, gone


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--write":
        write(sys.argv[2])
    elif len(sys.argv) == 3 and sys.argv[1] == "--check":
        ok, chw, ch, new, gone = check(sys.argv[2])
        if not ok:
            print("not comparable: the manifest was written under another compiler"); sys.exit(2)
        print("hardware-verified kernels changed: %d; other recorded kernels changed: %d; new: %d; gone: %d" % (len(chw), len(ch), len(new), len(gone)))
        for f, k in chw:
            print("  CHANGED (hw):", f, k)
        sys.exit(1 if chw else 0)
    else:
        print(__doc__)
