"""The rule oracle/__init__.py states, enforced: the oracle is the CHECKER.  Nothing under theseus_amd/ imports, names or loads
it (nor tests/, nor the reference), and in bench.py nothing that touches it runs between the clock's start and stop."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _imports(tree):
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield node.lineno, a.name
        elif isinstance(node, ast.ImportFrom):
            if node.level == 0 and node.module:
                yield node.lineno, node.module


def _py_files(top):
    for d, _, fs in os.walk(top):
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_package_never_imports_the_oracle():
    bad = []
    for path in _py_files(os.path.join(ROOT, "theseus_amd")):
        src = open(path).read()
        tree = ast.parse(src, path)
        rel = os.path.relpath(path, ROOT)
        for lineno, mod in _imports(tree):
            root = mod.split(".")[0]
            if root in ("oracle", "tests"):
                bad.append(f"{rel}:{lineno} imports {mod}")
        # dynamic routes: importlib / __import__ / a path under oracle/ in a string literal
        for node in ast.walk(tree):
            if isinstance(node, ast.Constant) and isinstance(node.value, str) and not _is_docstring(tree, node):
                v = node.value
                if v == "oracle" or v.startswith("oracle.") or v.startswith("oracle/") or "/oracle/" in v:
                    bad.append(f"{rel}:{node.lineno} names the oracle in a string: {v!r}")
    assert not bad, bad


def _is_docstring(tree, const):
    for node in ast.walk(tree):
        if isinstance(node, (ast.Module, ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)) and node.body:
            first = node.body[0]
            if isinstance(first, ast.Expr) and first.value is const:
                return True
    return False


def test_native_sources_do_not_reference_the_oracle():
    csrc = os.path.join(ROOT, "theseus_amd", "csrc")
    for f in os.listdir(csrc) + ["../../include/theseus_hip.h"]:
        text = open(os.path.join(csrc, f)).read()
        assert "oracle/" not in text and "#include \"oracle" not in text, f


def _oracle_users(tree):
    """Module-level functions / classes of bench.py whose body imports the oracle, closed over the functions that call them."""
    defs = {n.name: n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef))}
    users = {name for name, node in defs.items() if any(m.split(".")[0] == "oracle" for _, m in _imports(node))}
    changed = True
    while changed:
        changed = False
        for name, node in defs.items():
            if name in users or name in ("pg_run", "ba_run", "simple_run", "sparse_run", "small_batch_run", "main"):
                continue
            called = {n.id for n in ast.walk(node) if isinstance(n, ast.Name)}
            if called & users:
                users.add(name)
                changed = True
    return users, defs


def test_bench_uses_the_oracle_only_outside_the_timed_region():
    path = os.path.join(ROOT, "bench.py")
    tree = ast.parse(open(path).read(), path)
    assert not [m for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) for _, m in _imports(n) if m.split(".")[0] == "oracle"], \
        "bench.py must not import the oracle at module level"
    users, defs = _oracle_users(tree)
    assert {"cpu_baseline", "exact_reference", "oracle_implicit"} <= users
    checked = 0
    for fname in ("pg_run", "ba_run", "sparse_run", "small_batch_run"):
        fn = defs.get(fname)
        if fn is None:
            continue
        # a timed region: from ``t0 = time.perf_counter()`` to the next ``timer<...>.enabled = False``
        starts = [n.lineno for n in ast.walk(fn) if isinstance(n, ast.Assign) and isinstance(n.value, ast.Call)
                  and ast.unparse(n.value) == "time.perf_counter()" and ast.unparse(n.targets[0]) == "t0"]
        stops = [n.lineno for n in ast.walk(fn) if isinstance(n, ast.Assign) and ast.unparse(n.targets[0]).endswith(".enabled")
                 and ast.unparse(n.value) == "False"]
        assert starts and stops, fname
        for hi in stops:        # (a function may time several legs: each stop closes the region its nearest earlier start opened)
            lo = max(x for x in starts if x < hi)
            inside = [(n.lineno, n.id) for n in ast.walk(fn) if isinstance(n, ast.Name) and n.id in users and lo <= n.lineno <= hi]
            inside += [(ln, m) for ln, m in _imports(fn) if m.split(".")[0] == "oracle" and lo <= ln <= hi]
            assert not inside, (fname, lo, hi, inside)
            checked += 1
    assert checked >= 2


def test_entry_point_smoke_is_the_only_other_importer():
    """Outside tests/, oracle/, tools/ and bench.py only __graft_entry__.py (build(): "does the checker import"; smoke(): the
    check itself) may import the oracle."""
    allowed = {"bench.py", "__graft_entry__.py"}
    bad = []
    for f in os.listdir(ROOT):
        if f.endswith(".py") and f not in allowed:
            tree = ast.parse(open(os.path.join(ROOT, f)).read(), f)
            bad += [f"{f}:{ln}" for ln, m in _imports(tree) if m.split(".")[0] == "oracle"]
    for path in _py_files(os.path.join(ROOT, "examples")):
        tree = ast.parse(open(path).read(), path)
        bad += [f"{os.path.relpath(path, ROOT)}:{ln}" for ln, m in _imports(tree) if m.split(".")[0] == "oracle"]
    assert not bad, bad
