"""CPU-side checks of the C-ABI boundary: the shared library builds, loads and exports every symbol that
include/nanosim_amd.h declares; ctypes mirrors have the C layout.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from nanosim_amd import engine, model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nanosim_amd.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    return engine.load_library()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ns_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert set(names) == set(engine.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.ns_abi_version() == model.NS_ABI_VERSION


def test_struct_layouts_match_the_header():
    """sizeof/offsetof from a C compile of the header == the ctypes mirrors."""
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "nanosim_amd.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(ns_model_tables), sizeof(ns_params), sizeof(ns_batch_info),
         sizeof(ns_event), sizeof(ns_piece), sizeof(ns_read), sizeof(ns_hp_class));
  printf("%zu %zu %zu %zu %zu\n", offsetof(ns_model_tables, trans), offsetof(ns_model_tables, kde),
         offsetof(ns_model_tables, qual_thr), offsetof(ns_model_tables, hp), offsetof(ns_params, min_len));
  printf("%zu %zu %zu %zu %zu\n", sizeof(ns_cs_hist), offsetof(ns_cs_hist, dic), offsetof(ns_cs_hist, error_list),
         offsetof(ns_cs_hist, max_match), offsetof(ns_cs_hist, ms_kernel));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, src])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [int(x) for x in out]
    T = model.NsModelTables
    assert sizes[:7] == [C.sizeof(T), C.sizeof(model.NsParams), C.sizeof(model.NsBatchInfo),
                         model.EVENT_DTYPE.itemsize, model.PIECE_DTYPE.itemsize, model.READ_DTYPE.itemsize,
                         C.sizeof(model.NsHpClass)]
    assert sizes[7:12] == [T.trans.offset, T.kde.offset, T.qual_thr.offset, T.hp.offset, model.NsParams.min_len.offset]
    from nanosim_amd.characterize import NsCsHist as H
    assert sizes[12:] == [C.sizeof(H), H.dic.offset, H.error_list.offset, H.max_match.offset, H.ms_kernel.offset]


def test_no_gpu_means_loud_failure(lib):
    """Without a device ns_create must fail (never fall back to a CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.EngineError):
        engine.Engine(0)


def test_product_does_not_touch_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nanosim_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "ns_oracle" not in txt and "oracle_lib" not in txt and "libns_oracle" not in txt, f
