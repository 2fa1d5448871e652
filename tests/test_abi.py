"""The C-ABI library loads on a machine without a GPU and exports every symbol include/mcaller_hip.h declares."""
import ctypes
import os
import re

from tests import helpers as H


def test_library_exports_every_declared_symbol():
    from mcaller_amd import _lib
    header = open(os.path.join(H.REPO, 'include', 'mcaller_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    names = sorted(set(re.findall(r'\b(mc_[a-z_0-9]+)\s*\(', header)))
    assert len(names) >= 15
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    _lib.lib()
    assert b'mcaller_hip' in _lib.lib().mc_version()


def test_no_gpu_means_error_not_fallback():
    """Without a GPU the device entry points fail loudly (this test only asserts behaviour on GPU-less machines)."""
    from mcaller_amd import _lib
    L = _lib.lib()
    ctx = ctypes.c_void_p()
    rc = L.mc_ctx_create(0, ctypes.byref(ctx))
    if rc == 0:                       # a GPU is present: fine, clean up
        L.mc_ctx_destroy(ctx)
        return
    assert rc < 0 and b'no CPU fallback' in L.mc_last_error()


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(H.REPO, 'mcaller_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert 'oracle' not in re.sub(r'#.*', '', src).replace('"""', ''), f


def test_every_library_call_from_python_has_argtypes():
    """A ctypes call without argtypes passes a Python int as a C int: a byte offset beyond 2^31 arrives cut (mc_eventalign_read_cuts_at
    did, for a day).  Every mc_* function the package calls has its argument types declared, but for those that take none or small ints."""
    import glob
    import re
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mcaller_amd')
    src = ''.join(open(f).read() for f in glob.glob(os.path.join(here, '*.py')))
    called = set(re.findall(r'\b(?:lib\(\)|L|_lib\.lib\(\))\.(mc_[a-z0-9_]+)\(', src))
    typed = set(re.findall(r'L\.(mc_[a-z0-9_]+)\.argtypes', open(os.path.join(here, '_lib.py')).read()))
    assert called - typed <= {'mc_host_cores', 'mc_last_error', 'mc_repr_fixed4'}, sorted(called - typed)
