"""oracle/_ref front-end: the reference's own rasterizer and simple-knn sources, compiled for the host by
oracle/build_ref.py, behind the same Python interface as oracle/oracle.py.

TEST INFRASTRUCTURE ONLY (same rule as oracle.py: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).
Use: pin the restatement (tests/test_oracle_ref.py), generate golden fixtures (tests/golden/make_ref_fixtures.py),
and check the HIP path directly against the reference's code on the GPU box (the built library travels there;
/root/reference does not).
"""
import functools

from . import build_ref, oracle

_backend = None


def available():
    return build_ref.build() is not None


def lib():
    global _backend
    if _backend is None:
        path = build_ref.build()
        if path is None:
            raise RuntimeError("oracle/_ref is not built and /root/reference is not present")
        _backend = oracle.Backend(path, "ref_", False)
        _backend.set_threads = _backend.L.ref_set_threads
    return _backend


def set_threads(n):
    """n=1: blocks run in grid order on one thread, i.e. a fixed order of the reference's float atomicAdds; 0: all cores."""
    lib().set_threads(int(n))


def forward(*a, **k):
    return oracle.forward(*a, backend=lib(), **k)


backward = oracle.backward            # dispatches on the ForwardResult's backend


def mark_visible(*a):
    return oracle.mark_visible(*a, backend=lib())


def dist2(points):
    return oracle.dist2(points, backend=lib())
