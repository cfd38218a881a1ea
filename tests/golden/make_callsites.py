"""Collect every call the reference makes into its `qeft_cuda` extension, and the list the extension binds, as DATA.

Run in the build container (reads /root/reference as text; nothing is imported or executed):

    python tests/golden/make_callsites.py            # writes tests/golden/qeft_cuda_callsites.json

What is recorded (no source text, only names and counts):
  * `bound`: the names of `m.def("<name>", ...)` in qeft/kernel/qeft_cuda.cpp, with the `py::arg("<x>")` names and which of
    them carry a default;
  * `calls`: for each `qeft_cuda.<fn>(...)` call expression in the reference's Python (qlinear.py, monkeypatch/*.py): file,
    line, positional-argument count and keyword names;
  * `aliases`: `self.gemv = qeft_cuda.<fn>` style bindings (qlinear.py:223-234) together with the calls made through the
    alias (`self.gemv(...)`, `self.gemm(...)`): positional count and keyword names.
tests/test_shim_callsites.py checks the shim (`qeft_cuda` at the repo root) against this file everywhere, and against a fresh
parse wherever /root/reference exists.
"""
import ast
import json
import os
import re
import sys

REF = os.environ.get("QEFT_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qeft_cuda_callsites.json")
PY_FILES = ["qeft/qlinear.py", "qeft/monkeypatch/ftllama_modeling.py"]
CPP = "qeft/kernel/qeft_cuda.cpp"


def parse_bound(cpp_text):
    """m.def("name", &fn, "doc" [, py::arg("a") [= default], ...]); commented-out lines are skipped."""
    text = "\n".join(line.split("//")[0] for line in cpp_text.splitlines())
    bound = {}
    for m in re.finditer(r'm\.def\(\s*"(\w+)"(.*?)\);', text, re.S):
        name, rest = m.group(1), m.group(2)
        args = []
        for a in re.finditer(r'py::arg\("(\w+)"\)\s*(=)?', rest):
            args.append({"name": a.group(1), "has_default": a.group(2) is not None})
        bound[name] = args
    return bound


def _call_record(node, path):
    return {"file": path, "line": node.lineno, "n_positional": len(node.args),
            "keywords": [k.arg for k in node.keywords if k.arg is not None]}


def parse_python(path, text):
    tree = ast.parse(text)
    calls, aliases = [], {}
    for node in ast.walk(tree):
        # self.<alias> = qeft_cuda.<fn>
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Attribute) and isinstance(node.value.value, ast.Name) \
                and node.value.value.id == "qeft_cuda":
            for t in node.targets:
                if isinstance(t, ast.Attribute) and isinstance(t.value, ast.Name) and t.value.id == "self":
                    aliases.setdefault(t.attr, {"targets": [], "calls": []})
                    aliases[t.attr]["targets"].append({"fn": node.value.attr, "file": path, "line": node.lineno})
    for node in ast.walk(tree):
        if not isinstance(node, ast.Call) or not isinstance(node.func, ast.Attribute):
            continue
        f = node.func
        if isinstance(f.value, ast.Name) and f.value.id == "qeft_cuda":
            rec = _call_record(node, path)
            rec["fn"] = f.attr
            calls.append(rec)
        elif isinstance(f.value, ast.Name) and f.value.id == "self" and f.attr in aliases:
            aliases[f.attr]["calls"].append(_call_record(node, path))
    return calls, aliases


def collect(ref=REF):
    with open(os.path.join(ref, CPP)) as fh:
        bound = parse_bound(fh.read())
    calls, aliases = [], {}
    for p in PY_FILES:
        with open(os.path.join(ref, p)) as fh:
            c, a = parse_python(p, fh.read())
        calls += c
        for k, v in a.items():
            aliases.setdefault(k, {"targets": [], "calls": []})
            aliases[k]["targets"] += v["targets"]
            aliases[k]["calls"] += v["calls"]
    calls.sort(key=lambda r: (r["file"], r["line"]))
    return {"bound": bound, "calls": calls, "aliases": aliases}


if __name__ == "__main__":
    rec = collect()
    with open(OUT, "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    print(f"{OUT}: {len(rec['bound'])} bound functions, {len(rec['calls'])} direct calls, "
          f"{sum(len(v['calls']) for v in rec['aliases'].values())} calls through aliases", file=sys.stderr)
