"""The `cramjam/` shim and its hand-written stubs against the native module: every name a stub declares exists, and every keyword a
stub declares is accepted by the function's own argument parser (a TypeError would mean the stub and pymod.cpp disagree).
Reference surface: src/cramjam/lz4.pyi:51-152, snappy.pyi:33-94, __init__.pyi:28-130 (NOT copied: the stubs are written from
pymod.cpp's keyword lists).  Runs without a GPU: a call that gets past argument parsing fails with RuntimeError (no device)."""
import ast
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _defs(path):
    tree = ast.parse(open(path).read())
    funcs = [n for n in tree.body if isinstance(n, ast.FunctionDef)]
    classes = [n for n in tree.body if isinstance(n, ast.ClassDef)]
    return funcs, classes


def test_import_cramjam_is_the_engine():
    import cramjam
    import cramjam.lz4
    import cramjam_amd
    from cramjam.snappy import compress_raw
    assert cramjam.lz4.compress_block is cramjam_amd.lz4.compress_block
    assert compress_raw is cramjam_amd.snappy.compress_raw
    assert cramjam.Buffer is cramjam_amd.Buffer and cramjam.DecompressionError is cramjam_amd.DecompressionError
    assert os.path.exists(os.path.join(ROOT, "cramjam", "py.typed"))


@pytest.mark.parametrize("mod", ["lz4", "snappy"])
def test_stub_functions_exist_and_accept_their_keywords(mod):
    import cramjam
    m = getattr(cramjam, mod)
    funcs, classes = _defs(os.path.join(ROOT, "cramjam", mod + ".pyi"))
    assert {f.name for f in funcs} == {n for n in dir(m) if not n.startswith("_") and n not in ("Compressor", "Decompressor")}
    for f in funcs:
        fn = getattr(m, f.name)
        names = [a.arg for a in f.args.args]
        kwargs = {}
        for i, n in enumerate(names):
            kwargs[n] = b"\x04\x00\x00\x00\x40abcd" if i == 0 else (bytearray(64) if n == "output" else None)
        try:
            if len(names) == 1:
                fn(kwargs[names[0]])                       # METH_O functions take their argument positionally
            else:
                fn(**kwargs)
        except TypeError as e:                            # keyword / arity mismatch between stub and parser
            pytest.fail("%s.%s%r: %s" % (mod, f.name, tuple(names), e))
        except Exception:
            pass                                           # no device, or a codec error on the dummy bytes: parsing succeeded
    for c in classes:
        cls = getattr(m, c.name)
        for meth in (n for n in c.body if isinstance(n, ast.FunctionDef)):
            assert hasattr(cls, meth.name), (mod, c.name, meth.name)


def test_stub_classes_of_the_package():
    import cramjam
    funcs, classes = _defs(os.path.join(ROOT, "cramjam", "__init__.pyi"))
    for c in classes:
        cls = getattr(cramjam, c.name)
        for meth in (n for n in c.body if isinstance(n, ast.FunctionDef)):
            assert hasattr(cls, meth.name), (c.name, meth.name)
    b = cramjam.Buffer(data=b"abc", copy=True)
    assert b.read(n_bytes=2) == b"ab" and b.seek(position=0, whence=0) == 0 and b.len() == 3
