"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/dynhip.h declares; the ctypes table matches the header; creating a
context without a device fails loudly (there is no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "dynhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = re.findall(r"\b(?:int|void\*?|const char\*|dh_ctx\*)\s+\**(dh_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S)
    return {name: [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else []
            for name, args in protos}


def test_header_parses():
    fns = header_functions()
    assert len(fns) >= 25
    for must in ("dh_create", "dh_rebuild", "dh_rwalk_batch", "dh_slice_batch",
                 "dh_unif_batch", "dh_contains", "dh_bound_draw",
                 "dh_seed_children", "dh_scale_to_logvol"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from dynesty_amd import _lib
    path = _lib.lib_path()
    assert os.path.exists(path), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(path)
    for name in header_functions():
        assert hasattr(lib, name), f"{name} declared in dynhip.h but not exported"


def test_ctypes_table_matches_header():
    from dynesty_amd import _lib
    fns = header_functions()
    assert set(_lib.SIGNATURES) == set(fns), \
        set(_lib.SIGNATURES) ^ set(fns)
    for name, (_, argtypes) in _lib.SIGNATURES.items():
        assert len(argtypes) == len(fns[name]), (name, len(argtypes), fns[name])


def test_version_and_no_device_behaviour():
    from dynesty_amd import _lib
    lib = _lib.load()
    assert lib.dh_version() == 100
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        pytest.skip("a GPU is present")
    assert lib.dh_device_count() == 0
    with pytest.raises(_lib.DynHipError, match="no HIP device|no CPU fallback"):
        _lib.Context(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under dynesty_amd/ may import
    it (only tests/, __graft_entry__.smoke and bench.py's cpu_baseline leg)."""
    pkg = os.path.join(ROOT, "dynesty_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
    # bench.py: the oracle is the checker only -- imported inside the post-timed-region check
    # (Shard.verify) and the cpu_baseline leg, never at module level or in the timed code
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    allowed = {"verify", "cpu_baseline", "_cpu_walk_worker"}

    def visit(node, fn):
        for ch in ast.iter_child_nodes(node):
            name = ch.name if isinstance(ch, (ast.FunctionDef, ast.AsyncFunctionDef)) else fn
            if isinstance(ch, ast.ImportFrom) and (ch.module or "").split(".")[0] == "oracle":
                assert fn in allowed, f"bench.py imports the oracle in {fn!r}"
            if isinstance(ch, ast.Import):
                assert not any(a.name.split(".")[0] == "oracle" for a in ch.names) or fn in allowed
            visit(ch, name)
    visit(tree, "<module>")
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):src.index("def reference_logz_gate")]
    timed = main[main.index("sh.rebuild()  # frames must exist"):main.index("verified = None")]
    assert "oracle" not in timed and "verify" not in timed
