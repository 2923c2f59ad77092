"""TEST INFRASTRUCTURE: the ggml plug-in's HOST logic on a box without a GPU.  Builds ggml_amd/csrc/backend/*.cpp a second time — against tools/emul/shim_plugin (a host stand-in
for the HIP runtime API) and the whole-library emulation libcdna4_emul.so (the product's kernel sources compiled for the CPU) — into build/lib_emul/libggml-cdna4-emul.so, and
runs oracle/_ref/split_harness (ggml's public API, the unmodified reference's libggml-base / ggml-cpu) on it: graph walk, peepholes, the hand-off of quantized activations, NORM
chains that leave the image, the resident buffer type.  Needs the reference tree (headers) and oracle/_ref (the harness).
    python tools/emul/plugin_emul_check.py q4_K 256 512 96 shared"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import lib_emul_check as L  # noqa: E402

REF = "/root/reference"
BACKEND = os.path.join(ROOT, "ggml_amd", "csrc", "backend")
SRCS = ["ggml_cdna4_backend.cpp", "ggml_cdna4_ops.cpp", "ggml_cdna4_split.cpp"]
HARNESS = os.path.join(ROOT, "oracle", "_ref", "split_harness")


def available():
    return os.path.exists(os.path.join(REF, "src", "ggml-backend-impl.h")) and os.path.exists(HARNESS) and os.path.exists(L.CLANG)


import importlib.util as _ilu
import sys as _sys
_blm = _sys.modules.get("cdna4_emul_buildlock")                        # (one instance per process: its lock is re-entrant by a process-wide depth count)
if _blm is None:
    _bl = _ilu.spec_from_file_location("cdna4_emul_buildlock", __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "buildlock.py"))
    _blm = _ilu.module_from_spec(_bl); _sys.modules["cdna4_emul_buildlock"] = _blm; _bl.loader.exec_module(_blm)
_locked = _blm.locked          # (xdist workers share build/: one build at a time)


@_locked
def build():
    so = L.build_so()
    out = os.path.join(os.path.dirname(so), "libggml-cdna4-emul.so")
    deps = [os.path.join(BACKEND, f) for f in os.listdir(BACKEND)] + [os.path.join(HERE, "shim_plugin", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "ggml_cdna4.h"), so]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        cmd = [L.CLANG, "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-DGGML_BACKEND_DL", "-DGGML_BACKEND_BUILD", "-DGGML_BACKEND_SHARED", "-DGGML_SHARED",
               "-I" + os.path.join(HERE, "shim_plugin"), "-I" + os.path.join(REF, "include"), "-I" + os.path.join(REF, "src"), "-I" + os.path.join(ROOT, "include"), "-I" + BACKEND,
               "-o", out] + [os.path.join(BACKEND, s) for s in SRCS] + [so, "-Wl,-rpath," + os.path.dirname(so)]
        subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out


def harness(args, env=None, timeout=1800):
    """-> the JSON object of the harness's last output line (None: the environment cannot host the emulation)"""
    e = dict(os.environ, **{k: str(v) for k, v in (env or {}).items()})
    r = subprocess.run([HARNESS, build()] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, env=e)
    if "cannot host the emulation" in r.stderr:
        return None
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    print(json.dumps(harness(sys.argv[1:] or ["q4_K", "256", "512", "96", "shared"])))
