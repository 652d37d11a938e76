"""The SEXP-free half of the Rcpp shim (epialleler_amd/r/epihip_shim_core.hpp) compiled with g++ and driven from C++
(tests/cpp/test_shim_core.cpp): R and Rcpp are not in this image, so this is as close to the reference-side binding
as a test here can get.  The resident flow (upload once, threshold -> cx without re-upload, the fused report, lMHL)
is checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAM = os.path.join(H.GOLDEN, "bam", "capture.bam")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from epialleler_amd import _lib
    _lib.build()
    out = str(tmp_path_factory.mktemp("shim") / "test_shim_core")
    csrc = os.path.join(ROOT, "epialleler_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "epialleler_amd", "r"), os.path.join(ROOT, "tests", "cpp", "test_shim_core.cpp"),
                           "-o", out, "-L", csrc, "-lepihip", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                           "-lamdhip64", "-lpthread"])
    return out


def test_shim_core_host_side(exe):
    r = subprocess.run([exe, "cpu", BAM], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "shim core cpu ok: 2968 templates" in r.stdout


@pytest.mark.gpu
def test_shim_core_resident_flow(exe, tmp_path):
    r = subprocess.run([exe, "gpu", BAM, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    t = H.bam("capture.bam")
    ld = lambda name, dt: np.fromfile(os.path.join(str(tmp_path), name), dtype=dt)
    p = orc.threshold_reads(t["xm"], t["off"], "Z", "z", "XH", "xh", 2, 0.5, 0.1)
    assert np.array_equal(ld("pass.i32", np.int32), p)
    assert np.array_equal(ld("beta.f64", np.float64).view(np.uint64), orc.get_xm_beta(t["xm"], t["off"], "Z", "z").view(np.uint64))
    names = ("rname", "strand", "pos", "context", "meth", "unmeth")
    want = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], p, "Z")
    H.assert_reports_equal({k: ld("cx_%s.i32" % k, np.int32) for k in names}, want)
    assert want["pos"].size == 15408                                                   # test_generateCytosineReport.R:32-50
    want = orc.cx_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], None, "ZXH")
    H.assert_reports_equal({k: ld("cxall_%s.i32" % k, np.int32) for k in names}, want)
    wm = orc.mhl_report(t["xm"], t["off"], t["rname"], t["strand"], t["start"], "Zz", 0, 0, 0.1)
    got = {k: ld("mhl_%s.i32" % k, np.int32) for k in ("rname", "strand", "pos", "context", "coverage")}
    got["length"], got["lmhl"] = ld("mhl_length.f64", np.float64), ld("mhl_lmhl.f64", np.float64)
    H.assert_reports_equal(got, wm, float_cols=("length", "lmhl"))
