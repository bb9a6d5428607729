"""Import the real reference (dynesty 3.0.0 from /root/reference/py) where it
exists -- the build container only.  SURVEY.md appendix B: the package needs a
dist-info for its __version__ lookup."""
import os
import sys
import tempfile

# DYNESTY_REF_PY: another place to find the reference's `py/` directory (tools/tapb_hw.py stages a scratch copy,
# never committed, for a run on the GPU box)
REF = os.environ.get("DYNESTY_REF_PY", "/root/reference/py")


def have_reference():
    return os.path.isdir(os.path.join(REF, "dynesty"))


def import_reference():
    if "dynesty" in sys.modules:
        return sys.modules["dynesty"]
    if not have_reference():
        raise ImportError("reference not present")
    shim = tempfile.mkdtemp(prefix="dynesty_shim_")
    di = os.path.join(shim, "dynesty-3.0.0.dist-info")
    os.makedirs(di)
    with open(os.path.join(di, "METADATA"), "w") as f:
        f.write("Metadata-Version: 2.1\nName: dynesty\nVersion: 3.0.0\n")
    sys.path.insert(0, shim)
    sys.path.insert(0, REF)
    import dynesty
    return dynesty


class canonical_eigh:
    """Context manager: inside it every `lalg.eigh` call of dynesty.bounding returns eigenvectors with the
    device's sign convention (largest-magnitude component positive) -- the SURVEY section 7a-iii harness for
    comparing whole runs seed for seed (bounding.py:212, 261, 1338 are the call sites)."""

    def __enter__(self):
        import dynesty.bounding as db
        from oracle.bounding_ref import canon_cols
        self.db, self.orig = db, db.lalg
        orig = db.lalg

        class Proxy:
            def __getattr__(self, name):
                return getattr(orig, name)

            @staticmethod
            def eigh(*a, **kw):
                lam, vec = orig.eigh(*a, **kw)
                return lam, canon_cols(vec)
        db.lalg = Proxy()
        return self

    def __exit__(self, *exc):
        self.db.lalg = self.orig
        return False
