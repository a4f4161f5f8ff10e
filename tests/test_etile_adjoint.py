"""Reverse mode of the head algebra of the matrix-core gradient path (waveflow_amd/csrc/wf_etile_adjoint.h), checked on the CPU: the header is
scalar-generic, tests/etile_adjoint_check.cpp instantiates it in double precision with g++ and compares every pullback with central differences."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_head_pullbacks_match_central_differences(tmp_path):
    exe = tmp_path / "adjcheck"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "etile_adjoint_check.cpp")], check=True, cwd=ROOT)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    names = [line.split()[0] for line in r.stdout.splitlines() if line.strip()]
    assert {"r_triple", "jmul", "flow_head", "prior_head"} <= set(names), r.stdout
