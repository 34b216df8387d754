"""Import the *unmodified* reference (read-only, /root/reference) in a container that
lacks cvxpy / cvxpylayers / gctl / colorama / irsim.

Only used by tests/golden/make_golden.py (fixture generation, in the build container).
Nothing under tests/ that runs with `-m gpu`, nor smoke(), nor bench.py imports this:
/root/reference does not exist on the GPU box.

The missing third-party modules are replaced by *inert* stubs: every attribute, call,
operator and comparison on a stub returns another stub.  This lets the reference's
constructors run (robot.__init__, NRMP.__init__ builds the cvxpy problem symbolically
and asserts `prob.is_dcp(dpp=True)`, neupan/blocks/nrmp.py:302) while every piece of
*numeric* reference code -- point flow, DUNE, A/B/C linearisation, fa/fb -- executes
unmodified.  The one call that cannot run is `CvxpyLayer.__call__`
(neupan/blocks/nrmp.py:144); the golden generator substitutes the oracle QP there and
labels every such vector "reference code with substituted solver".
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


class _Inert:
    """An object on which everything works and returns another inert object."""

    shape = ()

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Inert()

    def __getitem__(self, k):
        return _Inert()

    def __iter__(self):
        return iter(())

    def __bool__(self):
        return True

    def __len__(self):
        return 0

    def _op(self, *a, **k):
        return _Inert()

    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = _op
    __matmul__ = __rmatmul__ = __truediv__ = __rtruediv__ = __pow__ = _op
    __neg__ = __pos__ = __abs__ = _op
    __le__ = __ge__ = __lt__ = __gt__ = _op
    __eq__ = _op
    __ne__ = _op
    __hash__ = object.__hash__

    @property
    def T(self):
        return _Inert()

    def to(self, *a, **k):
        return self


def _stub_module(name):
    m = types.ModuleType(name)
    m.__path__ = []  # behave as a package so `from x.y import z` resolves

    def _getattr(attr):
        if attr.startswith("__") and attr.endswith("__"):
            raise AttributeError(attr)
        return _Inert

    m.__getattr__ = _getattr
    return m


_STUBBED = ("cvxpy", "cvxpylayers", "cvxpylayers.torch", "gctl", "colorama", "irsim")


def real_solver_stack_available():
    """True if the reference's real solver path could run here (never, today)."""
    try:
        import cvxpy  # noqa: F401
        import cvxpylayers  # noqa: F401
        import diffcp  # noqa: F401
        import ecos  # noqa: F401
        return True
    except Exception:
        return False


def import_reference():
    """Return the reference `neupan` package imported from /root/reference."""
    if not real_solver_stack_available():
        for name in _STUBBED:
            if name not in sys.modules:
                sys.modules[name] = _stub_module(name)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import neupan  # noqa: E402  (the reference package)
    assert neupan.__file__.startswith(REFERENCE_ROOT), neupan.__file__
    return neupan
