"""The Julia extension (graphneuralnetworks.jl_amd/julia/GNNlibGnnmpExt.jl) cannot be executed here (no Julia in the image),
so its @ccall sites are checked mechanically against include/gnnmp.h: every called symbol must be declared, with the same
number of arguments, each of the matching C type class (pointer / int / int64_t / float) and the declared return type.
A wrong argument count or a swapped Cint / Int64 is exactly the defect that would otherwise only surface on a Julia host."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "graphneuralnetworks.jl_amd", "julia", "GNNlibGnnmpExt.jl")
HDR = os.path.join(ROOT, "include", "gnnmp.h")


def c_prototypes():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t|const char \*)\s*(gnnmp_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        kinds = []
        args = " ".join(args.split())
        if args not in ("void", ""):
            for a in args.split(","):
                a = a.strip()
                if "*" in a or a.startswith("gnnmp_stream_t"):
                    kinds.append("ptr")
                elif a.startswith("int64_t") or a.startswith("uint64_t"):
                    kinds.append("i64")
                elif a.startswith("int ") or a.startswith("int\t"):
                    kinds.append("i32")
                elif a.startswith("float"):
                    kinds.append("f32")
                else:
                    raise AssertionError(f"unclassified C parameter {a!r} of {name}")
        protos[name] = ({"int": "i32", "int64_t": "i64", "const char *": "ptr"}[ret], kinds)
    return protos


JL_KIND = {"Ptr{Cvoid}": "ptr", "Ptr{Ptr{Cvoid}}": "ptr", "Ptr{Int64}": "ptr", "Ptr{Int32}": "ptr", "Ptr{Cint}": "ptr", "Cstring": "ptr",
           "Cint": "i32", "Int64": "i64", "UInt64": "i64", "Cfloat": "f32"}


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


def julia_ccalls():
    src = open(JL).read()
    src = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    calls = []
    for m in re.finditer(r"@ccall\s+libgnnmp\.(gnnmp_\w+)\(", src):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        args = split_top(src[i:j - 1])
        ret = re.match(r"::(\w+(?:\{\w+\})?)", src[j:]).group(1)
        kinds = []
        for a in args:
            t = a.strip().rsplit("::", 1)[1].strip()
            assert t in JL_KIND, f"{m.group(1)}: unknown Julia FFI type {t!r}"
            kinds.append(JL_KIND[t])
        calls.append((m.group(1), kinds, JL_KIND[ret]))
    return calls


def test_every_ccall_matches_the_header():
    protos = c_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 20, "the binding calls at least the hot-path entry points"
    for name, kinds, ret in calls:
        assert name in protos, f"{name} is not declared in include/gnnmp.h"
        cret, ckinds = protos[name]
        assert ret == cret, f"{name}: return type {ret} vs {cret}"
        assert kinds == ckinds, f"{name}: Julia passes {kinds}, gnnmp.h declares {ckinds}"


def test_the_binding_covers_the_reference_seam():
    """the three propagate methods of GNNlib/ext/GNNlibAMDGPUExt.jl:13-32, the layer overloads, rrules for every wrapper that
    hides a @ccall from Zygote, and an identity-keyed plan cache"""
    src = open(JL).read()
    for f in ("copy_xj", "e_mul_xj", "w_mul_xj"):
        assert re.search(rf"function GNNlib\.propagate\(::typeof\({f}\)", src)
    for sym in ("GNNlib.gcn_conv", "GNNlib.gat_conv", "GNNlib.reduce_nodes", "GNNlib.softmax_edge_neighbors",
                "GNNlib.aggregate_neighbors"):
        assert sym + "(" in src
    for wrapped in ("fused_propagate", "dense", "gat_attention", "gat_attention_drop", "edge_softmax", "scatter_edges", "segment_pool",
                    "GNNGraphs._gather"):
        assert f"ChainRulesCore.rrule(::typeof({wrapped})" in src, f"no rrule for {wrapped}"
    assert "objectid(s), objectid(t), g.num_nodes, self_loops" in src      # the cache key (round 1 keyed on s alone)
    assert "WeakKeyDict" not in "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    assert "gnnmp_fused_conv_f32" in src and "gnnmp_gat_conv_grad2_f32" in src and "gnnmp_gat_conv_train_f32" in src


# ---- what a Julia parser / method table would have told us: checked statically ------------------------------------
def test_blocks_and_brackets_balance():
    """every function / if / for / while / let / do / begin / struct / module / try has its `end`, every bracket closes — one stray `end`
    and the north-star drop-in does not load"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from julia_static import block_balance
    opens, ends, stack, dsq, dpar = block_balance()
    assert dsq == 0 and dpar == 0
    assert opens == ends and not stack, f"{opens} block openers, {ends} ends; unclosed: {stack[-3:]}"
    assert opens > 50


def test_every_rrule_returns_one_tangent_per_argument():
    """ChainRulesCore's contract: the pullback returns a tangent for the function itself and for EVERY positional argument"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from julia_static import rrules
    rs = rrules()
    assert len(rs) >= 7
    for name, npos, counts in rs:
        assert counts, f"rrule of {name}: no `return NoTangent(), ...` found in its pullback"
        for c in counts:
            assert c == npos + 1, f"rrule of {name}: {npos} positional arguments but a pullback return with {c} tangents"


def test_extended_methods_exist_in_the_reference_with_that_arity():
    """every GNNlib.f / GNNGraphs.f the extension adds a method to is a function the reference defines, callable with the same number
    of positional arguments (tests/golden/julia_api_table.json, generated from /root/reference by tests/golden/make_julia_api_table.py:
    names and arities only)"""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from julia_static import extended_methods
    table = json.load(open(os.path.join(ROOT, "tests", "golden", "julia_api_table.json")))
    ms = extended_methods()
    assert len(ms) >= 10
    for pkg, name, (lo, hi) in ms:
        key = f"{pkg}.{name}"
        assert key in table and table[key], f"{key} is not defined by the reference"
        assert any(rlo <= hi and lo <= rhi for rlo, rhi in table[key]), f"{key}: extension arity {lo}..{hi}, reference {table[key]}"
    if os.path.isdir("/root/reference"):            # in the build container the table must be current
        import subprocess
        before = json.dumps(table, sort_keys=True)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "make_julia_api_table.py")], stdout=subprocess.DEVNULL)
        assert json.dumps(json.load(open(os.path.join(ROOT, "tests", "golden", "julia_api_table.json"))), sort_keys=True) == before


def test_every_reference_name_the_extension_uses_exists():
    """`using GNNlib: ...` / `using GNNGraphs: ...` lists and every qualified `GNNlib.x` / `GNNGraphs.x` in the extension name something
    the reference defines (tests/golden/julia_api_table.json "_names": names only, generated from /root/reference) — a misspelt import
    is an UndefVarError at load time, i.e. the drop-in does not load.  GNNlib does `using GNNGraphs` (GNNlib/src/GNNlib.jl:10), so a
    name qualified with GNNlib may be GNNGraphs'."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from julia_static import strip_comments_and_strings
    names = json.load(open(os.path.join(ROOT, "tests", "golden", "julia_api_table.json")))["_names"]
    src = strip_comments_and_strings(open(JL).read())
    known = {"GNNlib": set(names["GNNlib"]) | set(names["GNNGraphs"]), "GNNGraphs": set(names["GNNGraphs"])}
    used = set(re.findall(r"\b(GNNlib|GNNGraphs)\.([A-Za-z_]\w*!?)", src))
    for m in re.finditer(r"^\s*(?:using|import)\s+(GNNlib|GNNGraphs)\s*:\s*(.*)$", src, flags=re.M):
        used.update((m.group(1), n.strip()) for n in m.group(2).split(",") if n.strip())
    assert len(used) >= 15
    for pkg, name in sorted(used):
        if name == pkg:
            continue
        assert name in known[pkg], f"{pkg}.{name} is not defined by the reference"


def test_round3_entry_points_are_bound():
    src = open(JL).read()
    for sym in ("gnnmp_graphconv_chain_f32", "gnnmp_chain_jobs_create", "gnnmp_shard_by_size", "gnnmp_allgather_f32", "gnnmp_segment_bounds",
                "gnnmp_gat_conv_drop_f32", "gnnmp_gat_conv_grad_drop_f32", "gnnmp_attn_conv_drop_f32"):
        assert sym in src, sym
