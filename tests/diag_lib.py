"""The DIAGNOSTIC build of the HIP library (tests/_build/libhipnlp_diag.so, __graft_entry__.build(): the same device code objects, the host
pass compiled with -DHIPNLP_DIAG).  Its hipnlp_create honours the HIPNLP_* environment overrides — kernel variant (HIPNLP_WAVES,
HIPNLP_SEPARATE_REDUCE, HIPNLP_HESS_LAYOUT, HIPNLP_HESS_DIRECT), launch numbering (HIPNLP_DEBUG_SEQ0), A/B switches (HIPNLP_EARLY_STORE,
HIPNLP_CONST_JAC, HIPNLP_VARY_CHECK, HIPNLP_HESS_LAM_ZERO_COPY) and the stale-mapping simulation (HIPNLP_DEBUG_MISDIRECT_AUTO) — which the
SHIPPED library does not read at all.  TEST INFRASTRUCTURE: tests that need a particular kernel variant create their handle from this
build (`HipNlp(..., library=path)`); everything else, and every parity reference in those tests, runs on the product library."""
import contextlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG_SO = os.path.join(ROOT, "tests", "_build", "libhipnlp_diag.so")


def diag_library():
    if not os.path.exists(DIAG_SO):
        raise ImportError(DIAG_SO + " is missing: python -c 'import __graft_entry__ as g; g.build()'")
    return DIAG_SO


@contextlib.contextmanager
def diag_overrides(**env):
    """with diag_overrides(HIPNLP_WAVES=8) as lib: eng = HipNlp(..., library=lib) — the overrides are read by hipnlp_create / at first
    use, so the handle is created (and, for HIPNLP_DEBUG_MISDIRECT_AUTO, used) inside the block"""
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        assert k.startswith("HIPNLP_"), k
        os.environ[k] = str(v)
    try:
        yield diag_library()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
