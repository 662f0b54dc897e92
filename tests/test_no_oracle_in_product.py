"""The product package must never import, call or fall back to anything under oracle/ (or any CPU path)."""
import ast
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(root):
    for d, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_never_imports_oracle():
    bad = []
    for path in _py_files(os.path.join(REPO, "objectdetectionpl_amd")):
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            if any(n == "oracle" or n.startswith("oracle.") for n in names):
                bad.append(path)
    assert not bad, bad


def test_bench_uses_oracle_only_in_cpu_baseline():
    src = open(os.path.join(REPO, "bench.py")).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(node))
            assert uses == (node.name == "cpu_baseline"), node.name
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any(getattr(n, "module", "") and n.module.startswith("oracle") for n in top)


def test_missing_library_raises(tmp_path, monkeypatch):
    import pytest
    from objectdetectionpl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.OdplError):
        _lib.lib()
