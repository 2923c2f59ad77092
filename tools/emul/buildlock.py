"""(Loaded ONCE per process under the name cdna4_emul_buildlock — see the snippet in the *_check.py modules.)  One build of an emulation binary at a time across processes (pytest-xdist workers share build/ and tools/emul/): an exclusive flock around every build() of this directory."""
import contextlib
import fcntl
import functools
import os

_LOCK = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "build", ".emul_build.lock")


_depth = 0                                                             # re-entrant within a process: a build() that calls another build() (flock is per open file)


@contextlib.contextmanager
def build_lock():
    global _depth
    if _depth:
        _depth += 1
        try:
            yield
        finally:
            _depth -= 1
        return
    os.makedirs(os.path.dirname(_LOCK), exist_ok=True)
    with open(_LOCK, "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        _depth = 1
        try:
            yield
        finally:
            _depth = 0
            fcntl.flock(f, fcntl.LOCK_UN)


def locked(fn):
    @functools.wraps(fn)
    def wrapper(*a, **k):
        with build_lock():
            return fn(*a, **k)
    return wrapper
