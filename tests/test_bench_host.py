"""Host-side logic of bench.py that can be checked without a GPU: the optional legs' time budget (a leg that would start
after the budget says so instead of running) and that what they record serialises into the one JSON line."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(monkeypatch, budget):
    monkeypatch.setenv("BENCH_OPTIONAL_BUDGET_S", str(budget))
    sys.path.insert(0, ROOT)
    sys.modules.pop("bench", None)
    return importlib.import_module("bench")


def test_optional_legs_respect_the_time_budget(monkeypatch):
    b = _bench(monkeypatch, -1000)                  # the budget is already spent: nothing may start
    res = b.diagnostics()
    json.dumps(res)
    flat = []
    for v in res.values():
        flat += list(v.values()) if isinstance(v, dict) else [v]
    assert flat and all(v in ("not built",) or str(v).startswith("skipped: time budget") for v in flat), res


def test_experimental_variants_are_distinct_and_explicit(monkeypatch):
    b = _bench(monkeypatch, 240)
    vs = {b.FUSEQ_VARIANT, b.FUSEQ_PFW_VARIANT, b.FUSEQ_WBL2_VARIANT, b.FUSEQ_GRP_VARIANT}
    assert len(vs) == 4 and all(0 < v < 2 ** 31 and (v & 0xFFFF) == 4119 and (v >> 16) in (1024, 3072, 5120, 9216) for v in vs)
    assert 0 < b.optional_time_left() <= 240   # (the default is 200 s; the env var of this test sets 240)
