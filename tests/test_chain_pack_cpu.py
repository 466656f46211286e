"""csrc/chain_pack.h — the best-fit-decreasing job packing the device kernel of graph_chain2.hip runs per batch — on the CPU: the same
functions (plain C++ under g++, no HIP), driven sequentially by tests/c_harness/pack_check.cpp against a brute-force best fit decreasing:
every packing valid (member graphs whole, jobs filled from slot 0 without holes, <= 64 rows), exactly as many jobs as sequential best fit
decreasing, record count inside the kernel's LDS table — on config 5's size law, U{1..64}, one size only, every size, 200 random batches."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packing_algorithm_against_brute_force(tmp_path):
    exe = str(tmp_path / "pack_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "graphneuralnetworks.jl_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c_harness", "pack_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "all packings valid and as tight as best fit decreasing" in r.stdout
    assert "U{20..40} x 8192" in r.stdout
