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


def test_detqmc_driver_with_gpu_model_and_cdwU(tmp_path):
    """The reference's driver over the GPU model class at cdwU != 0: `cdwU` travels through ModelParamsDetSDW into the library, the
    configuration streams gain configs-l.{binary,text}stream (src/detsdwopdim.cpp:4967-4986, :5015-5037) and hold the discrete field of the
    same chain the Python view of the library walks for the same parameters.  (No comparison with the reference's CPU program here: its
    own chain at cdwU != 0 depends on the last bit of a determinant at every null proposal)"""
    from detqmc_amd import DetSDW, SDWParams
    conf = open(os.path.join(CASE, "simulation.conf")).read()
    conf = conf.replace("thermalization = 40", "thermalization = 4").replace("sweeps = 40", "sweeps = 4").replace("saveInterval = 20", "saveInterval = 2")
    conf = conf.replace("saveConfigurationStreamText = false", "saveConfigurationStreamText = true").replace("jkBlocks = 5", "jkBlocks = 2")
    conf = conf.replace("globalShift = true", "globalShift = false")
    conf += "\ncdwU = 0.5\n"
    (tmp_path / "simulation.conf").write_text(conf)
    log = _run(str(tmp_path))
    assert "Measurements finished" in log
    L, m, N = 4, 20, 16
    lb = np.fromfile(str(tmp_path / "configs-l.binarystream"), dtype=np.int32)
    nconf = lb.size // (N * m)
    assert nconf >= 1 and lb.size == nconf * N * m and set(np.unique(lb)) <= {-2, -1, 1, 2}
    lt = np.loadtxt(str(tmp_path / "configs-l.textstream"), dtype=np.int64)
    assert np.array_equal(lt.astype(np.int32), lb)
    pb = np.fromfile(str(tmp_path / "configs-phi.binarystream"))
    assert pb.size == nconf * N * m * 2
    meta = "".join(_header(str(tmp_path / "results.values")))
    assert "cdwU = 0.5" in meta
    # the stream headers of BOTH fields against the files the reference's CPU program wrote for this configuration file
    # (tests/golden/detqmc_run_o2_L4_cdw/expected, from oracle/_ref/detqmcsdwo2_ref; src/detsdwopdim.cpp:5040-5108)
    exp = os.path.join(ROOT, "tests", "golden", "detqmc_run_o2_L4_cdw", "expected")
    assert (tmp_path / "simulation.conf").read_text() == open(os.path.join(os.path.dirname(exp), "simulation.conf")).read()
    for fn in ("configs-l.infoheader", "configs-phi.infoheader"):
        assert _header(str(tmp_path / fn)) == _header(os.path.join(exp, fn)), fn
    for fn in ("configs-l.textstream", "configs-phi.textstream"):
        assert _header(str(tmp_path / fn)) == _header(os.path.join(exp, fn + ".header")), fn
    # the same parameters through the Python view: the first configuration is written behind the first measured sweep -- 4 thermalisation
    # sweeps, sweep(false), sweep(true) with measureInterval = 2 (DetQMC::run, src/detqmc.h:435-505)
    rep = DetSDW(SDWParams(opdim=2, L=4, beta=2.0, dtau=0.1, s=10, r=-0.5, c=1.0, u=1.0, lambda_=1.0, mu=-0.5, weakZflux=True, delaySteps=8, cdwU=0.5,   # c, u: the defaults of the reference option parser
                           fermionMeasurements=True, rngSeed=1020304050, simindex=0))
    for _ in range(4):
        rep.sweepThermalization()
    rep.sweep(False)
    rep.sweep(True)
    l = rep.cdwl
    want = np.array([l[k, iy * L + ix] for ix in range(L) for iy in range(L) for k in range(1, m + 1)], dtype=np.int32)
    assert np.array_equal(lb[:N * m], want), "driver and library walk different chains"
    rep.close()


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


# ------------------------------------------------------------------------------------------------
# the reference's REPLICA-EXCHANGE driver DetQMCPT<> (Boost.MPI, one rank per replica) over the GPU-backed model
# ------------------------------------------------------------------------------------------------
PT_EXE = os.path.join(ROOT, "oracle", "_ref", "detqmcptsdwgpu")
PT_CASE = os.path.join(ROOT, "tests", "golden", "detqmcpt_run_o2_L4")
MPIEXEC = "/opt/conda/bin/mpiexec"          # the MPICH of this image (same image on the GPU box)


def _mpilib():
    """the three MPICH libraries are reached through symlinks (oracle/ref_build/Makefile, target mpilib): /opt/conda/lib
    must not enter the library path, its older libstdc++ would shadow the one the ROCm runtime needs"""
    d = os.path.join(ROOT, "oracle", "_ref", "mpilib")
    os.makedirs(d, exist_ok=True)
    for lib in ("libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0", "libmkl_rt.so.1"):
        dst = os.path.join(d, lib)
        if not os.path.exists(dst):
            if os.path.islink(dst):
                os.remove(dst)
            os.symlink(os.path.join("/opt/conda/lib", lib), dst)


def _compare_pt_tree(workdir):
    """every file of tests/golden/detqmcpt_run_o2_L4/expected (written by the reference's CPU program `mpiexec -n 4 detqmcptsdwo2_ref`):
    metadata headers equal line by line, values to 1e-9, jackknife errors to 1e-5, configuration streams byte for byte"""
    exp = os.path.join(PT_CASE, "expected")
    nfiles = 0
    for dirpath, _, files in os.walk(exp):
        rel = os.path.relpath(dirpath, exp)
        for fn in sorted(files):
            want, got = os.path.join(dirpath, fn), os.path.join(workdir, rel, fn)
            assert os.path.exists(got), "the GPU-backed replica-exchange run did not write " + os.path.join(rel, fn)
            nfiles += 1
            if fn.endswith(".binarystream"):
                assert open(got, "rb").read() == open(want, "rb").read(), os.path.join(rel, fn) + ": field configurations differ"
                continue
            assert _header(got) == _header(want), os.path.join(rel, fn) + ": metadata header differs"
            if fn.endswith(".infoheader"):
                continue
            a, b = _numbers(got), _numbers(want)
            assert len(a) == len(b), fn
            for ra, rb in zip(a, b):
                assert len(ra) == len(rb), fn
                for j, (x, y) in enumerate(zip(ra, rb)):
                    try:
                        fx, fy = float(x), float(y)
                    except ValueError:
                        assert x == y, fn
                        continue
                    tol = 1e-5 if (fn.startswith("results") and j == len(ra) - 1) else 1e-9
                    if fn.endswith(".series"):
                        tol = 2e-6                      # six significant digits in the file
                    assert abs(fx - fy) <= tol * max(abs(fy), 1e-3), (rel, fn, ra, rb)
    assert nfiles == 4 * 16 + 3


def test_detqmcpt_driver_with_gpu_model_over_mpi_writes_the_reference_output_tree(tmp_path):
    """mpiexec -n 4 detqmcptsdwgpu: DetQMCPT<DetSDWGpu, ModelParamsDetSDW> -- the reference's own replicaExchangeStep,
    per-control-parameter observable handlers, exchange statistics and per-parameter configuration streams, every replica on
    the GPU (4 processes, one card) -- against the output tree the reference's CPU program (mpiexec -n 4 detqmcptsdwo2_ref)
    wrote for the same configuration file: p<cpi>_r<value>/{results*.values, *.series, configs-phi.binarystream},
    exchange-{parameters,acceptance,diffusion}.values."""
    assert os.path.exists(PT_EXE), "oracle/_ref/detqmcptsdwgpu missing: run `make -C oracle/ref_build detqmcptsdwgpu` in the build container"
    assert os.path.exists(MPIEXEC), "MPICH launcher of the image not found"
    _mpilib()
    shutil.copy(os.path.join(PT_CASE, "simulation.conf"), tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([MPIEXEC, "-n", "4", PT_EXE, "-c", "simulation.conf"], cwd=str(tmp_path), capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "Measurements finished" in out.stdout
    _compare_pt_tree(str(tmp_path))
    for p in range(4):
        assert os.path.exists(tmp_path / ("simulation.%d.state" % p))


def test_one_device_per_rank_without_enough_devices_fails_cleanly(tmp_path):
    """Multi-GPU readiness on a one-GPU box: `mpiexec -n 2 detqmcptsdwgpu` with DQMC_DEVICE_PER_RANK=1 asks for device 1, which this
    box does not have.  dqmc_create must answer DQMC_ENODEV, rank 1 must report it and abort the job (MPI_Abort through the
    reference driver's own error path) -- no hang, no partial output tree.  Reference behaviour for a rank / parameter mismatch:
    /root/reference/src/detqmcpt.h:285-289 (throws at start-up), :805-851 (all ranks stop together)."""
    import time
    assert os.path.exists(PT_EXE) and os.path.exists(MPIEXEC)
    _mpilib()
    conf = [l for l in open(os.path.join(PT_CASE, "simulation.conf")) if not l.startswith("rValues = -1.1") and not l.startswith("rValues = -1.0")]
    (tmp_path / "simulation.conf").write_text("".join(conf))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DQMC_DEVICE_PER_RANK="1")
    env.pop("DQMC_DEVICE", None)
    t0 = time.time()
    out = subprocess.run([MPIEXEC, "-n", "2", PT_EXE, "-c", "simulation.conf"], cwd=str(tmp_path), capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0, out.stdout[-2000:]
    assert "rank 1" in out.stderr and "device ordinal out of range" in out.stderr, out.stderr[-3000:]
    assert time.time() - t0 < 120
    assert not [d for d in os.listdir(tmp_path) if d.startswith("p0_") or d.startswith("p1_") or d.startswith("exchange-")]


@pytest.mark.parametrize("launch", ["one process, 4 replicas in one batch", "2 processes x 2 replicas (gloo)",
                                    "rccl: torch.distributed.run, 1 rank x 4 replicas, backend nccl"])
def test_python_replica_exchange_driver_writes_the_reference_output_tree(tmp_path, launch):
    """The repo's OWN replica-exchange driver -- scripts/run_pt.py over detqmc_amd/pt.py (replica_exchange_step, ObservableRouterPT,
    write_exchange_statistics, time series, per-parameter configuration streams) -- run on the reference's configuration file must
    write the tree the reference's `mpiexec -n 4 detqmcptsdwo2` wrote: same exchange decisions (DetQMCPT::replicaExchangeStep,
    /root/reference/src/detqmcpt.h:963-1118), same per-control-parameter accumulation and jackknife errors
    (src/mpiobservablehandlerpt.cpp:65-110, 176-215), same exchange-*.values (src/detqmcpt.h:596-660).  Once with all four replicas
    in one batch on one rank, once spread over two ranks (torch.distributed, gloo, both on the one GPU of the box), and once through
    the RCCL branch the multi-GPU runs use (backend "nccl", process group initialised with the device, DEVICE tensors through
    all_gather / broadcast, the exchange actions reduced on the GPU straight into the send tensor: detsdw_exchange_actions_device) --
    with the one rank a one-GPU box allows."""
    import sys
    shutil.copy(os.path.join(PT_CASE, "simulation.conf"), tmp_path)
    script = os.path.join(ROOT, "scripts", "run_pt.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if launch.startswith("one"):
        cmd = [sys.executable, script, "--conf", "simulation.conf", "--stabilisation", "svd"]
    elif launch.startswith("rccl"):
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), script, "--conf", "simulation.conf", "--backend", "nccl", "--check"]
    else:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), script, "--conf", "simulation.conf", "--backend", "gloo", "--one-device", "--check"]
    out = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "Measurements finished" in out.stdout
    if launch.startswith("rccl"):
        assert "exchange payload: device tensors, backend nccl" in out.stdout
    _compare_pt_tree(str(tmp_path))


# ------------------------------------------------------------------------------------------------
# BASELINE config 1: the reference's DetQMC<> driver over the GPU-backed Hubbard model
# ------------------------------------------------------------------------------------------------
def test_detqmc_driver_with_gpu_hubbard_model_writes_the_reference_output_tree(tmp_path):
    """detqmchubbardgpu = DetQMC<DetHubbardGpu, ModelParams<DetHubbard>> (oracle/ref_build/dethubbardgpu.h) with the reference's
    option parser (src/maindetqmchubbard.cpp), against the output of the reference's own detqmchubbard program for the same
    configuration file; then a resume from simulation.state (saveContents / loadContents through the boost archive)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "detqmchubbardgpu")
    case = os.path.join(ROOT, "tests", "golden", "detqmc_run_hubbard_L4")
    assert os.path.exists(exe), "oracle/_ref/detqmchubbardgpu missing: run `make -C oracle/ref_build detqmchubbardgpu` in the build container"
    _mpilib()
    d = os.path.join(ROOT, "oracle", "_ref", "mpilib", "libmkl_rt.so.1")
    if not os.path.exists(d):
        os.symlink("/opt/conda/lib/libmkl_rt.so.1", d)

    def run(workdir, *extra):
        out = subprocess.run([exe, "-c", "simulation.conf"] + list(extra), cwd=str(workdir), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        return out.stdout

    def compare(workdir, series_only=False):
        exp = os.path.join(case, "expected")
        for fn in sorted(os.listdir(exp)):
            if series_only and not fn.endswith(".series"):
                continue
            got = os.path.join(str(workdir), fn)
            assert os.path.exists(got), fn
            if not series_only:
                assert _header(got) == _header(os.path.join(exp, fn)), fn + ": metadata header differs"
            a, b = _numbers(got), _numbers(os.path.join(exp, fn))
            assert len(a) == len(b), fn
            for ra, rb in zip(a, b):
                for j, (x, y) in enumerate(zip(ra, rb)):
                    try:
                        fx, fy = float(x), float(y)
                    except ValueError:
                        assert x == y, fn
                        continue
                    tol = 1e-4 if (fn.startswith("results") and j == len(ra) - 1) else 1e-9
                    assert abs(fx - fy) <= tol * max(abs(fy), 1e-3), (fn, ra, rb)

    a = tmp_path / "full"
    a.mkdir()
    shutil.copy(os.path.join(case, "simulation.conf"), a)
    log = run(a)
    assert "Measurements finished" in log
    compare(a)
    b = tmp_path / "resumed"
    b.mkdir()
    shutil.copy(os.path.join(case, "simulation.conf"), b)
    run(b, "--sweeps", "20")
    log = run(b, "--sweeps", "40")
    assert "State of previous simulation has been loaded" in log
    compare(b, series_only=True)
