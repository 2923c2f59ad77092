import os
import sys
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("CDNA4_TESTS_ON_EMULATOR") == "1":
        # tests/test_gpu_tests_on_the_emulator.py: selected -m gpu tests, unchanged, against the whole-library CPU emulation (tests/emul_torch.py)
        import emul_torch
        emul_torch.activate()


def pytest_collection_modifyitems(config, items):
    """Order of the -m gpu run (the driver runs it with -x): what ran on hardware longest ago comes first, what is newest last.  The stock
    harness's MUL_MAT / MUL_MAT_ID sweeps exercise EVERY accepted weight type, including the ones added after the round's last hardware
    session, so they move behind the five-format suites — to the end, with tests/test_gpu_widening.py — instead of sorting with 'backend'."""
    def late(it):
        f = os.path.basename(str(it.fspath))
        if f == "test_gpu_backend_plugin.py" and it.name in ("test_stock_harness[MUL_MAT]", "test_stock_harness[MUL_MAT_ID]"):
            return 1
        if f == "test_gpu_widening.py" and it.name == "test_stock_harness_flash_attn_ext":      # its sweep now includes the quantized K / V cases
            return 1
        return 0
    items.sort(key=late)          # stable: everything else keeps its place
