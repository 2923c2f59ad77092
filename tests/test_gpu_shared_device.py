"""-m gpu: the DEFAULT mode of the library and the plug-in is safe on a device that is not exclusively owned (round 6).

* default (no GGML_CDNA4_OWNED_DEVICE): no route ever waits for a co-resident work-group — with "another tenant" holding half the CUs (ggml_cdna4_debug_occupy: work-groups
  that sit on 100 KB of LDS each, so that none of our 130-KB work-groups fits beside them) the headline product is simply computed on the CUs that are left: correct, no fault;
* owned device (opt-in) + the same tenant: the one-launch step's grid barrier cannot complete while half of its work-groups are not resident.  Either the tenant leaves in
  time (correct result) or the wait runs into its bound — then the tile is NaN AND ggml_cdna4_device_fault() says so, the next ggml_cdna4_mul_mat returns a non-zero status,
  and the library has switched itself to the non-waiting routes.  Never NaN with status 0.
Each case runs in a child process (the mode is read once; a fault demotes the library for the rest of the process)."""
import json
import os
import subprocess
import sys
import pytest
import refutil as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    import gpu_util
    from ggml_amd import native
    native.lib()
    return gpu_util


CODE = r"""
import ctypes as C, json, sys, time, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import refutil as R
from ggml_amd import native, ops
L = native.lib()
m, k, b = 4096, 4096, 512
tenant_ms, hold_ms = int(sys.argv[1]), int(sys.argv[2])
w = R.random_weights(R.Q4_K, m, k, seed=11)
a = ops.QTensor.from_host_bytes(R.Q4_K, k, m, w, device="cuda:0")
x = np.random.default_rng(12).uniform(-1, 1, (b, k)).astype(np.float32)
xd = torch.from_numpy(x).cuda()
out = {"route": L.ggml_cdna4_mul_mat_route(int(R.Q4_K), m, k, b)}
y_ref = ops.mul_mat(a, xd).cpu().numpy()                      # alone on the device
rows = np.random.default_rng(0).choice(m, 64, replace=False); rs = R.row_size(R.Q4_K, k)
yo = R.o_mul_mat(R.Q4_K, np.concatenate([w[r * rs:(r + 1) * rs] for r in rows]), x, 64, k)
out["alone_rel_l2"] = R.rel_l2(y_ref[:, rows], yo)
release = torch.zeros(16, dtype=torch.int32).pin_memory()
side = torch.cuda.Stream()
ncu = torch.cuda.get_device_properties(0).multi_processor_count
native.check(L.ggml_cdna4_debug_occupy(ncu // 2, 100, release.data_ptr(), tenant_ms, side.cuda_stream))
time.sleep(0.05)                                               # the tenant is resident
y = torch.full((b, m), 7.0, dtype=torch.float32, device="cuda")
nws = L.ggml_cdna4_mul_mat_workspace_size(int(R.Q4_K), k, b)
ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
t0 = time.time()
rc = L.ggml_cdna4_mul_mat(int(R.Q4_K), a.data.data_ptr(), a.row_bytes, xd.data_ptr(), k, y.data_ptr(), m, m, k, b, ws.data_ptr(), ws.numel(), 0, 0, 0, st)
out["rc_under_tenant"] = rc
ev = torch.cuda.Event(); ev.record()
while not ev.query() and time.time() - t0 < hold_ms / 1000.0: time.sleep(0.01)
out["finished_while_tenant_held_cus"] = bool(ev.query())
release[0] = 1                                                 # the tenant leaves
torch.cuda.synchronize()
out["seconds"] = round(time.time() - t0, 2)
yh = y.cpu().numpy()
out["finite"] = bool(np.isfinite(yh).all())
out["rel_l2_vs_alone"] = R.rel_l2(yh, y_ref) if out["finite"] else None
out["fault_peek"] = L.ggml_cdna4_device_fault(0)
rc2 = L.ggml_cdna4_mul_mat(int(R.Q4_K), a.data.data_ptr(), a.row_bytes, xd.data_ptr(), k, y.data_ptr(), m, m, k, b, ws.data_ptr(), ws.numel(), 0, 0, 0, st)
out["rc_next_call"] = rc2
out["next_call_error"] = L.ggml_cdna4_last_error().decode()[:120] if rc2 else ""
out["fault_after"] = L.ggml_cdna4_device_fault(0)
out["route_after"] = L.ggml_cdna4_mul_mat_route(int(R.Q4_K), m, k, b)
rc3 = L.ggml_cdna4_mul_mat(int(R.Q4_K), a.data.data_ptr(), a.row_bytes, xd.data_ptr(), k, y.data_ptr(), m, m, k, b, ws.data_ptr(), ws.numel(), 0, 0, 0, st)
torch.cuda.synchronize()
yh3 = y.cpu().numpy()
out["rc_third_call"] = rc3
out["third_call_rel_l2_vs_alone"] = R.rel_l2(yh3, y_ref) if np.isfinite(yh3).all() else None
print(json.dumps(out))
""" % (R.ROOT, os.path.join(R.ROOT, "tests"))


def _run(env_extra, tenant_ms, hold_ms):
    env = {k: v for k, v in os.environ.items() if k not in ("GGML_CDNA4_OWNED_DEVICE", "GGML_CDNA4_SHARED_DEVICE")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CODE, str(tenant_ms), str(hold_ms)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_default_mode_is_correct_beside_another_tenant(gu):
    rec = _run({}, tenant_ms=3000, hold_ms=2000)
    gu.report(test="shared_device_default_with_tenant", **rec)
    assert rec["route"] == 10 and rec["alone_rel_l2"] < 1e-3                    # quantizer + k_gemm_kq_t64 with the ticketed split: nobody waits
    assert rec["rc_under_tenant"] == 0 and rec["finished_while_tenant_held_cus"], rec      # computed on the CUs that were left, while the tenant was still there
    assert rec["finite"] and rec["rel_l2_vs_alone"] == 0.0, rec                  # the same sums in the same order: bit-identical to the run alone on the device
    assert rec["fault_peek"] == 0 and rec["rc_next_call"] == 0 and rec["third_call_rel_l2_vs_alone"] == 0.0, rec


def test_owned_mode_beside_a_tenant_is_correct_or_an_error_status_never_silent_nan(gu):
    # the tenant stays longer than the grid barrier waits (2^22 polls, a few seconds): the barrier gives up
    rec = _run({"GGML_CDNA4_OWNED_DEVICE": "1"}, tenant_ms=15000, hold_ms=12000)
    gu.report(test="owned_device_with_tenant", **rec)
    assert rec["route"] == 11, rec                                              # the one-launch step (its grid barrier needs every work-group resident)
    if rec["finite"]:
        # the tenant left (or the dispatcher found room) before the bound: a correct product and no fault
        assert rec["rel_l2_vs_alone"] is not None and rec["rel_l2_vs_alone"] < 2e-6 and rec["fault_peek"] == 0, rec
    else:
        # the bound was hit: NaN tiles — and the library SAYS so: the fault word is set, the next call returns an error status and launches nothing, the mode is demoted,
        # and the call after that takes the non-waiting route and is correct
        assert rec["fault_peek"] != 0, rec
        assert rec["rc_next_call"] != 0 and "co-resident" in rec["next_call_error"], rec
        assert rec["fault_after"] == 0 and rec["route_after"] == 10, rec
        assert rec["rc_third_call"] == 0 and rec["third_call_rel_l2_vs_alone"] is not None and rec["third_call_rel_l2_vs_alone"] < 2e-6, rec


CODE_MORE = r"""
import ctypes as C, json, sys, time, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import refutil as R
from ggml_amd import native, ops
L = native.lib()
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(5)
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
cases = {}
# (a) prefill-sized MUL_MAT_ID on the work queue: persistent work-groups, partial tiles summed by the last arriver
ne, nu, nt, m, k = 8, 2, 512, 1024, 1024
rb = R.row_size(R.Q4_K, k)
we = dev(R.random_weights(R.Q4_K, ne * m, k, seed=3)); xe = dev(rng.standard_normal((nt, nu, k)).astype(np.float32))
ids = dev(np.stack([rng.permutation(ne)[:nu] for _ in range(nt)]).astype(np.int32))
wse = torch.empty(L.ggml_cdna4_mul_mat_id_workspace_size(int(R.Q4_K), k, ne, nu, nu, nt), dtype=torch.uint8, device="cuda")
def moe(y):
    return L.ggml_cdna4_mul_mat_id(int(R.Q4_K), we.data_ptr(), rb, m * rb, xe.data_ptr(), k, nu * k, ids.data_ptr(), nu, y.data_ptr(), m, nu * m, m, k, ne, nu, nu, nt, wse.data_ptr(), wse.numel(), st)
cases["mul_mat_id_work_queue"] = (moe, (nt, nu, m))
# (b) a grid far below the chip: 32 tiles, K split four / eight ways, summed by the last arriver
def mk(t, m_, k_, b_, seed):
    a = ops.QTensor.from_host_bytes(t, k_, m_, R.random_weights(t, m_, k_, seed=seed), device="cuda:0")
    x = dev(rng.uniform(-1, 1, (b_, k_)).astype(np.float32))
    ws = torch.empty(L.ggml_cdna4_mul_mat_workspace_size(int(t), k_, b_), dtype=torch.uint8, device="cuda")
    def f(y): return L.ggml_cdna4_mul_mat(int(t), a.data.data_ptr(), a.row_bytes, x.data_ptr(), k_, y.data_ptr(), m_, m_, k_, b_, ws.data_ptr(), ws.numel(), 0, 0, 0, st)
    return f, (b_, m_), (a, x, ws)
keep = []
for name, t, m_, k_, b_ in (("deep_split_q4_K_4096x8192x96", R.Q4_K, 4096, 8192, 96), ("split_in_two_q4_K_4096x11008x512", R.Q4_K, 4096, 11008, 512), ("q8_0_8192x4096x512", R.Q8_0, 8192, 4096, 512)):
    f, shp, k3 = mk(t, m_, k_, b_, 7); keep.append(k3); cases[name] = (f, shp)
alone = {}
for name, (f, shp) in cases.items():
    y = torch.full(shp, 7.0, dtype=torch.float32, device="cuda"); assert f(y) == 0, L.ggml_cdna4_last_error(); torch.cuda.synchronize(); alone[name] = y.cpu().numpy()
release = torch.zeros(16, dtype=torch.int32).pin_memory(); side = torch.cuda.Stream()
ncu = torch.cuda.get_device_properties(0).multi_processor_count
native.check(L.ggml_cdna4_debug_occupy(ncu // 2, 100, release.data_ptr(), 20000, side.cuda_stream))
time.sleep(0.05)
out = {}
for name, (f, shp) in cases.items():
    y = torch.full(shp, 7.0, dtype=torch.float32, device="cuda")
    t0 = time.time(); rc = f(y); ev = torch.cuda.Event(); ev.record()
    while not ev.query() and time.time() - t0 < 5.0: time.sleep(0.005)
    done = bool(ev.query())
    out[name] = {"rc": rc, "finished_while_tenant_held_cus": done, "bit_identical_to_alone": bool(done and np.array_equal(y.cpu().numpy().view(np.uint32), alone[name].view(np.uint32)))}
release[0] = 1
torch.cuda.synchronize()
out["fault"] = L.ggml_cdna4_device_fault(0)
print(json.dumps(out))
""" % (R.ROOT, os.path.join(R.ROOT, "tests"))


def test_default_mode_every_split_and_the_work_queue_finish_beside_a_tenant(gu):
    """nothing the default mode launches waits for a co-resident work-group: with a tenant on half the CUs the work-queue MUL_MAT_ID (persistent work-groups, parked partial tiles),
    a four- / eight-way K split, the split in two of C3 and a Q8_0 prefill all finish WHILE the tenant is there, bit-identical to the run alone on the device, no fault"""
    env = {k: v for k, v in os.environ.items() if k not in ("GGML_CDNA4_OWNED_DEVICE", "GGML_CDNA4_SHARED_DEVICE")}
    r = subprocess.run([sys.executable, "-c", CODE_MORE], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    gu.report(test="shared_device_default_with_tenant_more_routes", **rec)
    assert rec.pop("fault") == 0
    for name, v in rec.items():
        assert v["rc"] == 0 and v["finished_while_tenant_held_cus"] and v["bit_identical_to_alone"], (name, v)


@pytest.mark.skipif(not os.path.exists(os.path.join(R.REF_DIR, "test-backend-ops")), reason="oracle/_ref not built")
def test_stock_mul_mat_sweep_in_the_default_mode(gu):
    """the unmodified reference harness on the plug-in WITHOUT GGML_CDNA4_OWNED_DEVICE: what an ordinary ggml application gets"""
    from ggml_amd import native
    env = {k: v for k, v in os.environ.items() if k not in ("GGML_CDNA4_OWNED_DEVICE", "GGML_CDNA4_SHARED_DEVICE")}
    env.update(GGML_BACKEND_PATH=native.BACKEND_PATH)
    r = subprocess.run([os.path.join(R.REF_DIR, "test-backend-ops"), "test", "-o", "MUL_MAT", "-b", "CDNA40"], env=env, capture_output=True, text=True, timeout=600)
    txt = r.stdout + r.stderr
    gu.report(test="stock_mul_mat_default_mode", rc=r.returncode, tail=txt[-300:])
    assert r.returncode == 0 and "FAIL" not in txt, txt[-2000:]
