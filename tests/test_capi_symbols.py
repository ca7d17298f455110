"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/epsm.h declares; argument validation works without touching a device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = set()
    for name in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not name.endswith(".h"):
            continue
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"\b(epsm_[a-z0-9_]+)\s*\(", text))
    return sorted(syms)


@pytest.fixture(scope="module")
def lib():
    from epsm_mitsuba3_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_declares_something():
    syms = declared_symbols()
    assert "epsm_manifold_grad" in syms and "epsm_trace_paths" in syms and len(syms) >= 10


def test_every_declared_symbol_is_exported(lib):
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/epsm.h but not exported"


def test_abi_version_and_counts(lib):
    assert lib.epsm_abi_version() == 7
    assert lib.epsm_num_param_grads(0, 5) == 25
    assert lib.epsm_num_param_grads(1, 5) == 23


def test_argument_validation_without_device(lib):
    from epsm_mitsuba3_amd.records import EpsmVertexRecord
    recs = (EpsmVertexRecord * 1)()
    rc = lib.epsm_manifold_grad(0, 16, 9, None, C.addressof(recs), None, 4, 2, None, 0.1, None, None, None, None)
    assert rc == -22
    assert b"K must be" in lib.epsm_last_error()
    rc = lib.epsm_manifold_grad(7, 16, 2, None, C.addressof(recs), None, 4, 2, None, 0.1, None, None, None, None)
    assert rc == -22


def test_vertex_record_layout_matches_header():
    from epsm_mitsuba3_amd.records import EpsmVertexRecord
    text = open(os.path.join(ROOT, "include", "epsm.h")).read()
    body = re.search(r"typedef struct EpsmVertexRecord \{(.*?)\} EpsmVertexRecord;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^const\s+\w+\s*", "", decl)
        names += [n.strip(" *") for n in decl.split(",")]
    assert names == [f[0] for f in EpsmVertexRecord._fields_]
    assert C.sizeof(EpsmVertexRecord) == 8 * len(names)


def test_trace_structs_match_header_sizes():
    """ctypes mirrors of include/epsm_trace.h keep the C layout (compiled with gcc and compared)."""
    import subprocess, tempfile, textwrap
    from epsm_mitsuba3_amd import scene as S
    src = textwrap.dedent("""
        #include <stdio.h>
        #include "epsm_trace.h"
        int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(EpsmMesh), sizeof(EpsmBsdf), sizeof(EpsmEmitter),
                                sizeof(EpsmBvhNode), sizeof(EpsmSensor), sizeof(EpsmScene), sizeof(EpsmRecordOut)); return 0; }
    """)
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(td, "t"), os.path.join(td, "t.c")], check=True)
        sizes = [int(x) for x in subprocess.run([os.path.join(td, "t")], capture_output=True, text=True, check=True).stdout.split()]
    mine = [C.sizeof(S.EpsmMesh), C.sizeof(S.EpsmBsdf), C.sizeof(S.EpsmEmitter), 128, C.sizeof(S.EpsmSensor),
            C.sizeof(S.EpsmSceneC), C.sizeof(S.EpsmRecordOut)]
    assert sizes == mine, (sizes, mine)


def test_cxx_host_driver_builds_and_reports_missing_device():
    """examples/epsm_host_driver.cpp compiles against include/epsm.h, links the product library and, without
    a GPU, fails loudly at its device check (exit status 1) instead of computing anything on the host."""
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "-s"], check=True)
    exe = os.path.join(ROOT, "examples", "build", "epsm_host_driver")
    assert os.path.isfile(exe)
    import torch
    if torch.cuda.device_count() == 0:
        r = subprocess.run([exe, "1000", "2"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "no HIP device" in r.stderr


def test_launch_options_round_trip_without_a_device(lib):
    """include/epsm.h, epsm_set_option / epsm_get_option: process-wide launch options of the fused entry points; their initial
    values come from the environment when the library is loaded (no entry point calls getenv), unknown options and negative
    values are refused with -EINVAL."""
    from epsm_mitsuba3_amd import _lib
    small, rep = lib.epsm_get_option(_lib.OPT_SMALL_WAVEFRONT_PATHS), lib.epsm_get_option(_lib.OPT_REPLICAS)
    assert small >= 0 and rep in (0, 1)
    with _lib.options(small_wavefront_paths=12345, replicas=False):
        assert lib.epsm_get_option(_lib.OPT_SMALL_WAVEFRONT_PATHS) == 12345 and lib.epsm_get_option(_lib.OPT_REPLICAS) == 0
    assert lib.epsm_get_option(_lib.OPT_SMALL_WAVEFRONT_PATHS) == small and lib.epsm_get_option(_lib.OPT_REPLICAS) == rep
    assert lib.epsm_set_option(7, 1) == -22 and b"epsm_set_option" in lib.epsm_last_error()
    assert lib.epsm_set_option(_lib.OPT_REPLICAS, -1) == -22
    assert lib.epsm_get_option(99) == -1
