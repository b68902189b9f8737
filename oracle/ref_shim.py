"""Import the reference's Part 1 / Part 2 modules from /root/reference (THIS CONTAINER ONLY).

TEST INFRASTRUCTURE - never imported by the product.  Used by oracle/gen_golden.py to
produce the fixtures under tests/golden/ and by nothing else; /root/reference does not
exist on the GPU box.

The reference imports four packages that are not installed in this image
(SURVEY.md section 8c): ``numba`` (orderGenome.py:11-12), ``hmmlearn``
(scaffoldToChromosomes.py:10), ``community`` (scaffoldToChromosomes.py:14) and ``xarray``
(plotContactMaps.py:8).  None of them is on the hyperGeom=True / hmm=False / modularity=0
path except ``numba.jit``, for which the reference keeps a pure-Python twin of the same
loop (orderGenome.py:323-330).  We register inert stand-ins so the modules import:
``numba.jit`` returns the function unchanged, plotting is a no-op.  Nothing is installed
or fetched.
"""
import os
import sys
import types

REFERENCE_DIR = "/root/reference/HIC_ASSEMBLER"


def _identity_jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


def install():
    if not os.path.isdir(REFERENCE_DIR):
        raise RuntimeError("reference tree not present: fixtures can only be generated in the build container")
    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")
        nb.jit = _identity_jit
        sys.modules["numba"] = nb
    if "hmmlearn" not in sys.modules:
        hl = types.ModuleType("hmmlearn")
        hl.hmm = types.ModuleType("hmmlearn.hmm")
        sys.modules["hmmlearn"] = hl
        sys.modules["hmmlearn.hmm"] = hl.hmm
    if "community" not in sys.modules:
        sys.modules["community"] = types.ModuleType("community")
    if "plotContactMaps" not in sys.modules:
        pm = types.ModuleType("plotContactMaps")
        pm.plotContactMap = lambda *a, **k: None
        sys.modules["plotContactMaps"] = pm
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)


def load():
    """Return (scaffoldToChromosomes, orderGenome) reference modules."""
    install()
    import scaffoldToChromosomes as s2c      # noqa: E402  (reference module)
    import orderGenome as og                 # noqa: E402  (reference module)
    return s2c, og
