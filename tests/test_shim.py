"""shim/scg_shim.cpp (the Rcpp glue of INTEGRATION.md) type-checks against include/scg.h, and binds every
file-level entry point.  R is absent from this image, so the check runs against the type-level stand-in
tests/fake_rcpp/Rcpp.h; nothing is linked or executed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shim", "scg_shim.cpp")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_shim_type_checks():
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror",
                           "-I" + os.path.join(ROOT, "tests", "fake_rcpp"), "-I" + os.path.join(ROOT, "include"), SHIM])


def test_shim_binds_every_file_level_entry():
    header = open(os.path.join(ROOT, "include", "scg.h")).read()
    text = open(SHIM).read()
    file_level = sorted(set(re.findall(r"\bint (scg_(?:count_[a-z_]+barcodes[a-z_]*|match_barcodes))\(", header)))
    assert len(file_level) >= 9
    for name in file_level:
        assert re.search(r"\b" + name + r"\(", text), f"{name} is not bound by shim/scg_shim.cpp"


def test_shim_keeps_the_reference_signatures():
    """Exported names and arities of src/RcppExports.cpp:136-145 (fixed .Call registration)."""
    text = open(SHIM).read()
    arity = {"count_combo_barcodes_paired": 13, "count_combo_barcodes_single": 7, "count_dual_barcodes": 14,
             "count_dual_barcodes_single_end": 8, "count_random_barcodes": 6, "count_single_barcodes": 7, "match_barcodes": 4}
    for name, n in arity.items():
        m = re.search(r"//\[\[Rcpp::export\(rng=false\)\]\]\s*\nRcpp::List " + name + r"\(([^)]*)\)", text)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip()]) == n, name
