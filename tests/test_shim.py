"""The Velox-side adapter (shim/*.cpp). There is no Velox to compile against in this repository
(folly / fmt / glog / xsimd are absent), so the shim is compiled against tests/velox_api_stub - the
slice of the Velox API it touches, with the reference's signatures - and
 - type-checked (g++ -fsyntax-only -Wall -Wextra -Werror),
 - run on the CPU through the reference's own TPC-H Q1 plan (tests/cpp/shim_plan_test.cpp: count(0),
   avg's ROW(DOUBLE, BIGINT) intermediate in and out, the FilterProject -> HashAggregation fusion),
 - run on the GPU as operators of a Driver (tests/cpp/shim_operator_test.cpp)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = [os.path.join(ROOT, "shim", f) for f in ("Vx355Adapter.cpp", "Vx355JoinAdapter.cpp")]
INCLUDES = ["-I", os.path.join(ROOT, "tests", "velox_api_stub"), "-I", os.path.join(ROOT, "include"),
            "-I", os.path.join(ROOT, "shim"), "-I", os.path.join(ROOT, "tests", "cpp")]
LIBDIR = os.path.join(ROOT, "velox_amd")


def _build(test_source, out):
    from velox_amd import build
    build.build_lib()
    cmd = (["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror"] + INCLUDES +
           [os.path.join(ROOT, "tests", "cpp", test_source)] + SHIM +
           ["-L", LIBDIR, "-lvx355", f"-Wl,-rpath,{LIBDIR}", "-pthread", "-o", out])
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_shim_sources_type_check_against_the_velox_api_stub():
    for src in SHIM:
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror"] + INCLUDES + [src],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_shim_includes_only_headers_the_reference_has():
    """Every "velox/..." header the shim includes exists under that path in the reference (checked
    against a recorded list: /root/reference is not on the GPU box) and in the stub."""
    import re
    recorded = {
        "velox/common/future/VeloxPromise.h", "velox/common/memory/MemoryArbitrator.h", "velox/core/Expressions.h", "velox/core/PlanNode.h",
        "velox/core/QueryConfig.h", "velox/exec/Aggregate.h", "velox/exec/Driver.h", "velox/exec/FilterProject.h",
        "velox/exec/HashAggregation.h", "velox/exec/HashBuild.h", "velox/exec/HashProbe.h", "velox/exec/HashTable.h", "velox/exec/Operator.h", "velox/type/Filter.h",
        "velox/exec/OperatorUtils.h", "velox/exec/Task.h", "velox/vector/ComplexVector.h",
        "velox/vector/DecodedVector.h", "velox/vector/FlatVector.h"}
    used = set()
    for f in os.listdir(os.path.join(ROOT, "shim")):
        if f.endswith((".cpp", ".h")):
            used |= set(re.findall(r'#include "(velox/[^"]+)"', open(os.path.join(ROOT, "shim", f)).read()))
    assert used <= recorded, used - recorded
    for h in used:
        assert os.path.exists(os.path.join(ROOT, "tests", "velox_api_stub", h)), h
        if os.path.isdir("/root/reference/velox"):
            assert os.path.exists(os.path.join("/root/reference", h)), h


def test_stub_names_exist_in_the_reference():
    """tools/stub_drift.py: every class and function the API stub declares is a name of the reference's headers
    (the stub's own stand-ins are listed there with what they stand in for). Needs /root/reference."""
    if not os.path.isdir("/root/reference/velox"):
        pytest.skip("the reference tree is not on this machine")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stub_drift.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 not found" in r.stdout


def test_reference_q1_plan_is_accepted_by_the_shim(tmp_path):
    exe = _build("shim_plan_test.cpp", str(tmp_path / "shim_plan_test"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "the reference's Q1 plan is accepted" in r.stdout


@pytest.mark.gpu
def test_shim_operators_run_the_q1_plan_and_a_join_on_the_gpu(tmp_path):
    exe = _build("shim_operator_test.cpp", str(tmp_path / "shim_operator_test"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "results match the expected values" in r.stdout
