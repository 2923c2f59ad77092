import os
import sys
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (no GPU in the box) spreads over three pytest-xdist workers when the plug-in is installed and nobody asked for a worker count: its long tests are CPU
    emulations of kernels in child processes (12 min serially, 7 on three workers).  Never with a GPU: the -m gpu tests share ONE device and some kernels wait for
    co-resident work-groups of their own launch.  CDNA4_TESTS_SERIAL=1 keeps one process."""
    if os.environ.get("CDNA4_TESTS_SERIAL") or os.environ.get("PYTEST_XDIST_WORKER") or os.environ.get("CDNA4_TESTS_ON_EMULATOR") == "1":
        return None
    if getattr(config.option, "numprocesses", "absent") is not None:      # no xdist, or -n given
        return None
    try:
        import torch
        if torch.cuda.is_available():
            return None
    except Exception:  # noqa: BLE001
        return None
    config.option.numprocesses = 3
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The suite OWNS its device (one process at a time on a box of its own), and most of its route pins, hand-off and one-launch tests were written for the routes an
    # owned device takes: it says so, like bench.py does.  The DEFAULT mode (round 6: shared — nothing ever waits for a co-resident work-group) is what
    # tests/test_gpu_shared_device.py and test_abi.py::test_route_table_for_a_256_cu_part run, in child processes without this variable.
    os.environ.setdefault("GGML_CDNA4_OWNED_DEVICE", "1")
    if os.environ.get("CDNA4_TESTS_ON_EMULATOR") == "1":
        # tests/test_gpu_tests_on_the_emulator.py: selected -m gpu tests, unchanged, against the whole-library CPU emulation (tests/emul_torch.py)
        import emul_torch
        emul_torch.activate()


def pytest_collection_modifyitems(config, items):
    """Order of the -m gpu run: file order, except that the slow stock FLASH_ATTN_EXT sweep runs last.  The stock MUL_MAT / MUL_MAT_ID sweeps of the
    unmodified reference harness stay where they are, among the five-format suites (tests/test_gpu_backend_plugin.py): they are the strongest
    boundary evidence of the hot path and must not hide behind newer code (round 2 sorted them last and lost them to a -x stop)."""
    def late(it):
        f = os.path.basename(str(it.fspath))
        return 1 if (f == "test_gpu_widening.py" and it.name == "test_stock_harness_flash_attn_ext") else 0
    items.sort(key=late)          # stable: everything else keeps its place
