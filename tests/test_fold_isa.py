"""Build-time check of the ISA contract behind the in-kernel fold of split rows (csrc/common.h: coh_store / coh_load / coh_publish,
csrc/csr_reduce.h: chunk_arrive).  The partials of a split row cross workgroups (and XCDs, whose L2s are not coherent) through relaxed
agent-scope atomics + `s_waitcnt vmcnt(0)` — outside the HIP memory model, correct on gfx942 / gfx950 because of what those compile to.
This test disassembles the device code of the built objects and pins exactly that:
  * every FOLD instance of csr_rows_kernel / gat_fused_rows_kernel contains loads AND stores with the `sc1` bit and an agent-scope atomic add;
  * none of them contains a whole-L2 write-back / invalidate (`buffer_wbl2`, `buffer_inv`): a __threadfence() creeping back in would be
    correct but cost 170 us on the arxiv shape (LABNOTES.md §3), and its absence is what the hand-written ordering has to make up for;
  * the device code was built for gfx950 (common.h refuses to compile for anything but gfx942 / gfx950).
No GPU needed: hipcc cross-compiles, llvm-objdump reads the code object.  (VERDICT r5 item 7 / ADVICE r5.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDIR = os.path.join(ROOT, "graphneuralnetworks.jl_amd", "lib", "obj")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def device_functions(obj_name, tmp_path):
    """{demangled kernel name: disassembly text} of the gfx950 code object bundled in lib/obj/<obj_name>"""
    src = os.path.join(OBJDIR, obj_name)
    if not os.path.exists(src):
        import __graft_entry__ as ge
        ge.build()
    obj = os.path.join(tmp_path, obj_name)
    shutil.copy(src, obj)
    subprocess.check_call([OBJDUMP, "--offloading", obj], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    code = [f for f in os.listdir(tmp_path) if f.startswith(obj_name + ".") and "amdgcn" in f]
    assert len(code) == 1 and code[0].endswith("gfx950"), f"expected one gfx950 code object in {obj_name}, found {code}"
    asm = subprocess.check_output([OBJDUMP, "-d", os.path.join(tmp_path, code[0])], text=True)
    funcs, name, buf = {}, None, []
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name:
                funcs[name] = "\n".join(buf)
            name, buf = m.group(1), []
        elif name:
            buf.append(line)
    if name:
        funcs[name] = "\n".join(buf)
    names = list(funcs)
    dem = subprocess.check_output(["c++filt"], input="\n".join(names), text=True).splitlines()
    return {d: funcs[n] for n, d in zip(names, dem)}


def fold_instances(funcs, kernel):
    """the instances of `kernel` whose LAST template argument (FOLD) is true"""
    out = {}
    for name, text in funcs.items():
        m = re.search(r"gnnmp::" + kernel + r"<([^>]*)>\(", name)
        if m and m.group(1).split(",")[-1].strip() == "true":
            out[name] = text
    return out


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump in this image")
@pytest.mark.parametrize("obj,kernel", [("propagate.o", "csr_rows_kernel"), ("gat_fused.o", "gat_fused_rows_kernel")])
def test_fold_kernels_use_sc1_accesses_and_no_l2_maintenance(obj, kernel, tmp_path):
    funcs = device_functions(obj, str(tmp_path))
    fold = fold_instances(funcs, kernel)
    assert fold, f"no FOLD instance of {kernel} in {obj}"
    for name, text in fold.items():
        loads = [l for l in text.splitlines() if re.search(r"\bglobal_load_\w+", l) and "sc1" in l]
        stores = [l for l in text.splitlines() if re.search(r"\bglobal_store_\w+", l) and "sc1" in l]
        atomics = [l for l in text.splitlines() if re.search(r"\bglobal_atomic_add\w*", l)]
        assert loads, f"{name}: no sc1 load — the partials would be read from a (non-coherent) L2"
        assert stores, f"{name}: no sc1 store — the partials would stay in the writer's L2"
        assert atomics, f"{name}: no arrival-counter atomic"
        assert "buffer_wbl2" not in text and "buffer_inv" not in text, f"{name}: whole-L2 write-back / invalidate inside a FOLD kernel"
        # the publish: a full vmcnt(0) wait exists between the partial's stores and the counter's atomic (program order in the listing)
        assert re.search(r"s_waitcnt vmcnt\(0\)", text), f"{name}: no s_waitcnt vmcnt(0)"
    # and the non-FOLD instances must not pay for it
    plain = {n: t for n, t in funcs.items() if re.search(r"gnnmp::" + kernel + r"<", n) and n not in fold}
    assert plain
    for name, text in plain.items():
        assert "buffer_wbl2" not in text and "buffer_inv" not in text, name


def test_device_code_refuses_other_architectures():
    """common.h carries the compile-time guard: the fold's ordering is an ISA property of gfx942 / gfx950"""
    txt = open(os.path.join(ROOT, "graphneuralnetworks.jl_amd", "csrc", "common.h")).read()
    assert re.search(r"#if defined\(__HIP_DEVICE_COMPILE__\) && !\(defined\(__gfx942__\) \|\| defined\(__gfx950__\)\)\s*\n#error", txt)
