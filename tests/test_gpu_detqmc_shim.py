"""End-to-end check of the drop-in boundary: the reference's OWN driver template DetQMC<Model, ModelParams>
(src/detqmc.h) instantiated with the GPU-backed model DetSDWGpu (oracle/ref_build/detsdwgpu.h) and linked against
libdetqmc_amd.so -- oracle/_ref/detqmcsdwgpu, built by `make -C oracle/ref_build detqmcsdwgpu` in the build container
(it travels to the GPU box like the other prebuilt files).  It reads the reference's configuration file format and
must write the output tree the reference's CPU program wrote for the same file
(tests/golden/detqmc_run_o2_L4/expected, produced by oracle/_ref/detqmcsdwo2_ref = src/maindetqmcsdwo2.cpp).
Nothing here reads /root/reference."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "detqmcsdwgpu")
CASE = os.path.join(ROOT, "tests", "golden", "detqmc_run_o2_L4")


def _numbers(path):
    """data lines (not starting with #) as rows of floats / strings"""
    rows = []
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        rows.append(line.split())
    return rows


def _header(path):
    return [l for l in open(path) if l.startswith("#")]


def _run(workdir, *extra):
    assert os.path.exists(EXE), "oracle/_ref/detqmcsdwgpu missing: run __graft_entry__.build() in the build container"
    out = subprocess.run([EXE, "-c", "simulation.conf"] + list(extra), cwd=workdir, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return out.stdout


def _compare_tree(workdir, value_tol=1e-9, error_tol=1e-5):
    exp = os.path.join(CASE, "expected")
    for fn in sorted(os.listdir(exp)):
        got = os.path.join(workdir, fn)
        assert os.path.exists(got), "the GPU-backed run did not write " + fn
        if fn.endswith(".binarystream"):
            assert open(got, "rb").read() == open(os.path.join(exp, fn), "rb").read(), fn + ": field configurations differ"
            continue
        assert _header(got) == _header(os.path.join(exp, fn)), fn + ": metadata header differs"
        if fn.endswith(".infoheader"):
            continue
        a, b = _numbers(got), _numbers(os.path.join(exp, fn))
        assert len(a) == len(b), fn
        for ra, rb in zip(a, b):
            assert len(ra) == len(rb), fn
            for j, (x, y) in enumerate(zip(ra, rb)):
                try:
                    fx, fy = float(x), float(y)
                except ValueError:
                    assert x == y, fn
                    continue
                # results*.values: key, value, jackknife error (the error is a difference of nearly equal block means)
                tol = error_tol if (fn.startswith("results") and j == len(ra) - 1) else value_tol
                assert abs(fx - fy) <= tol * max(abs(fy), 1e-3), (fn, ra, rb)


def test_detqmc_driver_with_gpu_model_writes_the_reference_output_tree(tmp_path):
    shutil.copy(os.path.join(CASE, "simulation.conf"), tmp_path)
    log = _run(str(tmp_path))
    assert "Thermalization finished" in log and "Measurements finished" in log
    _compare_tree(str(tmp_path))
    assert os.path.exists(tmp_path / "simulation.state") and os.path.exists(tmp_path / "info.dat")


def test_detqmc_driver_resumes_from_its_state_file(tmp_path):
    """DetQMC::saveState / the resume constructor (src/detqmc.h:121-135, 266-325) carry the replica through
    DetSDWGpu::saveContents / loadContents: 40 + 20 sweeps, then a second process continues to 40 measurement
    sweeps -- time series and configuration stream must equal the uninterrupted run's."""
    shutil.copy(os.path.join(CASE, "simulation.conf"), tmp_path)
    _run(str(tmp_path), "--sweeps", "20")
    log = _run(str(tmp_path), "--sweeps", "40")
    assert "will resume simulation" in log and "State of previous simulation has been loaded" in log
    exp = os.path.join(CASE, "expected")
    assert open(tmp_path / "configs-phi.binarystream", "rb").read() == open(os.path.join(exp, "configs-phi.binarystream"), "rb").read()
    for fn in sorted(os.listdir(exp)):
        if not fn.endswith(".series"):
            continue
        a = np.array([float(r[0]) for r in _numbers(tmp_path / fn)])
        b = np.array([float(r[0]) for r in _numbers(os.path.join(exp, fn))])
        assert a.shape == b.shape, fn
        assert np.all(np.abs(a - b) <= 1e-9 * np.maximum(np.abs(b), 1e-3)), fn
