"""CPU-side checks of the drop-in boundary: libvx355.so builds for gfx950, loads
without a GPU and exports every symbol include/vx355.h declares. No compute is
launched here."""
import ctypes as C
import os
import re

import pytest

from velox_amd import abi, build, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_lib()
    return ops.lib()


def test_header_and_symbol_list_agree():
    text = open(os.path.join(ROOT, "include", "vx355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(vx355_[a-z0-9_]+)\s*\(", text))
    assert declared == set(ops.SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in ops.SYMBOLS:
        assert hasattr(lib, name), name


def test_calls_that_need_no_gpu(lib):
    assert lib.vx355_abi_version() == 9
    if lib.vx355_device_count() == 0:
        # No device: init must fail loudly, and operators refuse to be created.
        assert lib.vx355_init(0) != abi.OK
        assert lib.vx355_last_error()
        spec, keep = ops.make_agg_spec([0], [abi.BIGINT], [(abi.AGG_COUNT_STAR, -1, abi.BIGINT)],
                                       abi.STEP_SINGLE)
        h = C.c_void_p()
        assert lib.vx355_agg_create(C.byref(spec), C.byref(h)) == abi.EINVAL
        assert b"vx355_init" in lib.vx355_last_error()


def test_product_package_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "velox_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in src and "oracle_lib" not in src, f
                assert not re.search(r'#include\s+"[^"]*oracle', src), f


def test_ctypes_structures_have_the_layout_of_the_header(tmp_path):
    """The ctypes mirrors in velox_amd/abi.py against the C compiler's view of include/vx355.h:
    size of every struct and offset of every field (a field added on one side only, or in another
    order, would otherwise corrupt arguments silently)."""
    import subprocess
    pairs = {"vx355_column": abi.Column, "vx355_batch": abi.Batch, "vx355_out_column": abi.OutColumn,
             "vx355_value_id_spec": abi.ValueIdSpec, "vx355_agg_fn": abi.AggFn, "vx355_agg_spec": abi.AggSpec,
             "vx355_agg_stats": abi.AggStats, "vx355_key_filter": abi.KeyFilter,
             "vx355_join_build_spec": abi.JoinBuildSpec, "vx355_join_table_stats": abi.JoinTableStats,
             "vx355_join_probe_spec": abi.JoinProbeSpec, "vx355_filter_term": abi.FilterTerm,
             "vx355_join_filter_term": abi.JoinFilterTerm, "vx355_factor": abi.Factor,
             "vx355_projection": abi.Projection}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vx355.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for field, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{field} %zu\\n", offsetof({cname}, {field}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(out[cname]) == C.sizeof(cls), cname
        for field, _ in cls._fields_:
            assert int(out[f"{cname}.{field}"]) == getattr(cls, field).offset, f"{cname}.{field}"


def test_shim_sources_only_use_declared_entry_points():
    """shim/Vx355Adapter.{h,cpp} and shim/Vx355JoinAdapter.{h,cpp} are compiled on the Velox side (no
    Velox here); what can be checked without it: every vx355_* function, type and constant the adapters
    use exists in include/vx355.h, and every source is in the CMake fragment."""
    header = open(os.path.join(ROOT, "include", "vx355.h")).read()
    declared = set(re.findall(r"\b(vx355_[a-z0-9_]+|VX355_[A-Z0-9_]+)\b", header))
    used = set()
    for name in ("Vx355Adapter.h", "Vx355Adapter.cpp", "Vx355JoinAdapter.h", "Vx355JoinAdapter.cpp"):
        text = open(os.path.join(ROOT, "shim", name)).read()
        text = re.sub(r"//[^\n]*", "", text)
        used |= set(re.findall(r"\b(vx355_[a-z0-9_]+|VX355_[A-Z0-9_]+)\b", text))
    used -= {"vx355_adapter"}   # (a CMake target name, not an ABI symbol)
    assert used and used <= declared, sorted(used - declared)
    cmake = open(os.path.join(ROOT, "shim", "CMakeLists.txt")).read()
    assert "Vx355Adapter.cpp" in cmake and "Vx355JoinAdapter.cpp" in cmake
