"""The drop-in boundary from C and C++: include/vx355.h must be plain C, include/vx355.hpp (the
header-only mirror of exec::Operator's interface) must compile against it, and a C++ program
written like the reference's operator tests must get the right answers through libvx355.so."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "velox_amd")
SRC = os.path.join(ROOT, "tests", "cpp", "operator_test.cpp")


def _compile(out):
    from velox_amd import build
    build.build_lib()
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-I", INCLUDE, SRC, "-L", LIBDIR, "-lvx355",
           f"-Wl,-rpath,{LIBDIR}", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_header_is_plain_c_and_cpp_consumer_compiles(tmp_path):
    assert shutil.which("gcc") and shutil.which("g++")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c",
                        os.path.join(INCLUDE, "vx355.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    _compile(str(tmp_path / "operator_test"))


@pytest.mark.gpu
def test_cpp_operator_program_runs_on_the_gpu(tmp_path):
    exe = _compile(str(tmp_path / "operator_test"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "match the expected results" in r.stdout
