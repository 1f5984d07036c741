"""bench.py and tools/*.py only run on the GPU box: a name that is never bound (a refactoring slip) would surface there, at round end.  This is a
small static check in the spirit of pyflakes' undefined-name rule: every name a function loads must be a parameter, bound somewhere in the
function (or an enclosing one), a module-level name, or a builtin."""
import ast
import builtins
import glob
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + sorted(
    glob.glob(os.path.join(ROOT, "halo2-lib_amd", "*.py")))


def _bound_names(node):
    """names bound directly in this scope (not in nested function / class scopes, whose own names are collected separately)"""
    out = set()

    def visit(n, top):
        if not top and isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef, ast.Lambda)):
            if hasattr(n, "name"):
                out.add(n.name)
            return
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                out.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            out.update(n.names)
        elif isinstance(n, ast.arg):
            out.add(n.arg)
        for c in ast.iter_child_nodes(n):
            visit(c, False)

    visit(node, True)
    return out


def _check_scope(node, outer, problems, path):
    scope = outer | _bound_names(node)
    if isinstance(node, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
        pass
    for n in ast.iter_child_nodes(node):
        _walk(n, scope, problems, path)


def _walk(n, scope, problems, path):
    if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
        for d in getattr(n, "decorator_list", []):
            _walk(d, scope, problems, path)
        _check_scope(n, scope, problems, path)
        return
    if isinstance(n, ast.ClassDef):
        _check_scope(n, scope, problems, path)
        return
    if isinstance(n, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
        _check_scope(n, scope, problems, path)
        return
    if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in scope:
        problems.append("%s:%d: name %r is never bound" % (os.path.relpath(path, ROOT), n.lineno, n.id))
    for c in ast.iter_child_nodes(n):
        _walk(c, scope, problems, path)


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(p, ROOT) for p in FILES])
def test_no_unbound_names(path):
    tree = ast.parse(open(path).read(), path)
    problems = []
    _check_scope(tree, set(dir(builtins)) | {"__file__", "__name__", "__doc__"}, problems, path)
    assert not problems, "\n".join(problems)


def test_recall_table_is_current():
    """the [UPSTREAM-RECALL] table names the product and oracle lines to change per recalled statement; its locations are
    found by anchor strings (tools/recall_table.py) and must match the committed text"""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "recall_table.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
