#!/usr/bin/env python
"""Generator of tests/golden/julia_api_table.json: for every GNNlib / GNNGraphs function the Julia extension adds methods to, the
positional arities the REFERENCE defines (min..max over default arguments), scanned from /root/reference's sources.  Only names and
integers are stored — no reference source text.  Run in the build container (the reference is not on the GPU box):
    python tests/golden/make_julia_api_table.py"""
import json
import os
import re
import sys

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from julia_static import extended_methods, split_top, strip_comments_and_strings   # noqa: E402

PKG_DIRS = {"GNNlib": ["GNNlib/src", "GNNlib/ext"], "GNNGraphs": ["GNNGraphs/src", "GNNGraphs/ext"]}


def definitions(pkg, name):
    """positional arities (as [min, max] pairs) of every method `name` defined in the package's sources"""
    out = set()
    pat = re.compile(r"(?:^|\s)(?:function\s+)?(?:[A-Za-z_][\w.]*\.)?" + re.escape(name) + r"\(", re.M)
    for d in PKG_DIRS[pkg]:
        for base, _, files in os.walk(os.path.join(REF, d)):
            for f in files:
                if not f.endswith(".jl"):
                    continue
                src = strip_comments_and_strings(open(os.path.join(base, f)).read())
                for m in pat.finditer(src):
                    i = m.end()
                    depth, j = 1, i
                    while depth and j < len(src):
                        depth += {"(": 1, ")": -1}.get(src[j], 0)
                        j += 1
                    tail = src[j:j + 40].lstrip()
                    head = src[max(0, m.start() - 12):m.end()]
                    is_def = "function" in head or re.match(r"(where\s*\{[^}]*\}\s*)?=(?!=)", tail) is not None
                    if not is_def:
                        continue
                    args = split_top(src[i:j - 1].split(";")[0]) if src[i:j - 1].strip() else []
                    npos = len(args)
                    ndef = sum(1 for a in args if re.search(r"[^=!<>]=[^=]", a))
                    if any(a.strip().endswith("...") for a in args):
                        out.add((npos - 1, 99))
                    else:
                        out.add((npos - ndef, npos))
    return sorted(out)


def defined_names(pkg):
    """every top-level name the package's sources define (functions, structs, abstract types, constants / type aliases) — names only"""
    names = set()
    pats = [re.compile(r"^\s*(?:@\w+\s+)*function\s+(?:[A-Za-z_][\w.]*\.)?([A-Za-z_]\w*!?)\s*[({]", re.M),
            re.compile(r"^(?:[A-Za-z_][\w.]*\.)?([A-Za-z_]\w*!?)\([^\n=]*\)\s*(?:where\s*\{[^}]*\}\s*)?=(?!=)", re.M),
            re.compile(r"^\s*(?:mutable\s+)?struct\s+([A-Za-z_]\w*)", re.M),
            re.compile(r"^\s*abstract\s+type\s+([A-Za-z_]\w*)", re.M),
            re.compile(r"^\s*const\s+([A-Za-z_]\w*)\s*=", re.M)]
    for d in PKG_DIRS[pkg]:
        for base, _, files in os.walk(os.path.join(REF, d)):
            for f in files:
                if f.endswith(".jl"):
                    src = strip_comments_and_strings(open(os.path.join(base, f)).read())
                    for pat in pats:
                        names.update(pat.findall(src))
    return sorted(names)


def main():
    table = {}
    for pkg, name, _arity in extended_methods():
        table[f"{pkg}.{name}"] = definitions(pkg, name)
    table["_names"] = {pkg: defined_names(pkg) for pkg in PKG_DIRS}
    path = os.path.join(ROOT, "tests", "golden", "julia_api_table.json")
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    print(path, {k: v for k, v in table.items() if k != "_names"}, {k: len(v) for k, v in table["_names"].items()})


if __name__ == "__main__":
    main()
