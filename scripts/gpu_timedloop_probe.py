"""what the driver's `bench.py --steps 20 --warmup 5` loop measures beside the kernel: the same timed loop (barrier + synchronize, 5 warm-up steps, 20 timed steps) after an idle
pause, with and without >= 50 ms of untimed steps in front of it (clock ramp), next to the HIP-event time of the same launches in steady state."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
M, K, Bn = B.HEAD
w, x, how = B.prescribed(B.Q4_K, M, K, 0, M, Bn)
h = B.Hot(dev, B.Q4_K, w, M, K, x)
def timed(preheat_ms, W=5, Kk=20):
    if preheat_ms:
        t = time.perf_counter()
        while time.perf_counter() - t < preheat_ms * 1e-3:
            for _ in range(8): h.step()
            torch.cuda.synchronize()
    for _ in range(W): h.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(Kk): h.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / Kk * 1e6
out = {"events_us_steady": round(B.events_us(h.step, 200, 10), 2)}
for pre in (0, 50, 0, 50, 0, 50):
    time.sleep(0.5)
    out.setdefault("timed20_us_preheat_%d" % pre, []).append(round(timed(pre), 2))
time.sleep(0.5)
out["timed200_w20_us_no_preheat"] = round(timed(0, 20, 200), 2)
print(json.dumps(out))
