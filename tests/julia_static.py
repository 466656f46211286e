"""Static reading of graphneuralnetworks.jl_amd/julia/GNNlibGnnmpExt.jl (no Julia in the image): shared by the tests and by the
generator of tests/golden/julia_api_table.json."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "graphneuralnetworks.jl_amd", "julia", "GNNlibGnnmpExt.jl")


def strip_comments_and_strings(src):
    """Julia source with comments removed and string literals emptied (so that keywords / brackets inside them do not count)"""
    out, i, n = [], 0, len(src)
    while i < n:
        ch = src[i]
        if ch == "#":
            if src.startswith("#=", i):
                j = src.find("=#", i + 2)
                i = n if j < 0 else j + 2
                continue
            j = src.find("\n", i)
            i = n if j < 0 else j
            continue
        if ch == '"':
            if src.startswith('"""', i):
                j = src.find('"""', i + 3)
                out.append('""')
                i = n if j < 0 else j + 3
                continue
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('""')
            i = j + 1
            continue
        if ch == "'" and i + 2 < n and (src[i + 2] == "'" or (src[i + 1] == "\\" and i + 3 < n and src[i + 3] == "'")):
            j = src.find("'", i + 2)
            out.append("' '")
            i = j + 1
            continue
        out.append(ch)
        i += 1
    return "".join(out)


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _call_args(src, i):
    """(positional argument strings, index after the closing parenthesis) of the call whose '(' is at i - 1"""
    depth, j = 1, i
    while depth:
        depth += {"(": 1, ")": -1}.get(src[j], 0)
        j += 1
    inner = src[i:j - 1]
    # keyword arguments follow the first top-level ';'
    d, cut = 0, len(inner)
    for k, ch in enumerate(inner):
        if ch in "([{":
            d += 1
        elif ch in ")]}":
            d -= 1
        elif ch == ";" and d == 0:
            cut = k
            break
    return split_top(inner[:cut]), j


def extended_methods(src=None):
    """[(package, function, (min_arity, max_arity))] for every `GNNlib.f(...)` / `GNNGraphs.f(...)` method the extension defines"""
    src = strip_comments_and_strings(open(JL).read()) if src is None else src
    out = []
    for m in re.finditer(r"(?:^|\n)\s*(function\s+)?(GNNlib|GNNGraphs)\.(\w+)\(", src):
        args, j = _call_args(src, m.end())
        tail = src[j:j + 60].lstrip()
        if not m.group(1) and not re.match(r"(where\s*\{[^}]*\}\s*)?=(?!=)", tail):
            continue                                                     # a call, not a definition
        ndef = sum(1 for a in args if re.search(r"[^=!<>]=[^=]", a))
        out.append((m.group(2), m.group(3), (len(args) - ndef, len(args))))
    return out


def rrules(src=None):
    """[(wrapped function, positional argument count, [tangent counts of every `return` in its pullback])]"""
    src = strip_comments_and_strings(open(JL).read()) if src is None else src
    out = []
    for m in re.finditer(r"function ChainRulesCore\.rrule\(::typeof\(([\w.]+)\)\s*,?", src):
        args, j = _call_args(src, src.index("(", m.start()) + 1)
        npos = len(args) - 1                                             # without ::typeof(f) itself
        # the pullback: the first nested `function ..._pullback(` after the signature
        pm = re.search(r"function (\w*pullback)\(", src[j:])
        assert pm, f"rrule of {m.group(1)} has no pullback function"
        body_start = j + pm.end()
        body_end = src.index(f"return y, {pm.group(1)}", body_start) if f"return y, {pm.group(1)}" in src[body_start:] else None
        if body_end is None:
            mm = re.search(r"return \w+, " + pm.group(1), src[body_start:])
            assert mm, f"rrule of {m.group(1)} does not return its pullback"
            body_end = body_start + mm.start()
        counts = []
        for r in re.finditer(r"\breturn\s+(NoTangent\(\)[^\n]*)", src[body_start:body_end]):
            counts.append(len(split_top(r.group(1))))
        out.append((m.group(1), npos, counts))
    return out


BLOCK_OPENERS = {"function", "if", "for", "while", "let", "do", "begin", "struct", "module", "try", "macro", "quote"}


def block_balance(src=None):
    """(openers, ends, trace) counted outside brackets: inside [...] `end` is an index and `for` / `if` belong to a comprehension;
    inside (...) they belong to a generator or are part of an expression like (end - 1)"""
    src = strip_comments_and_strings(open(JL).read()) if src is None else src
    depth_sq = depth_par = 0
    opens = ends = 0
    stack = []
    for m in re.finditer(r"[\[\]()]|[A-Za-z_@][\w!]*", src):
        tok = m.group(0)
        if tok == "[":
            depth_sq += 1
        elif tok == "]":
            depth_sq -= 1
        elif tok == "(":
            depth_par += 1
        elif tok == ")":
            depth_par -= 1
        elif depth_sq == 0 and depth_par == 0:
            if tok in BLOCK_OPENERS:
                opens += 1
                stack.append((tok, src.count("\n", 0, m.start()) + 1))
            elif tok == "end":
                ends += 1
                assert stack, f"`end` without an opener at line {src.count(chr(10), 0, m.start()) + 1}"
                stack.pop()
        assert depth_sq >= 0 and depth_par >= 0, f"unbalanced bracket at line {src.count(chr(10), 0, m.start()) + 1}"
    return opens, ends, stack, depth_sq, depth_par
