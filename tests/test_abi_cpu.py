"""CPU-only checks of the drop-in boundary: the shared library builds/loads, exports every symbol the
headers declare, the host-side logic that needs no GPU (RNG restatement, parameter checks, exchange
probability) matches the reference fixtures, and the product fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def lib():
    import detqmc_amd
    if not os.path.exists(detqmc_amd.LIB_PATH):
        from detqmc_amd.build import build
        build(verbose=False)
    return detqmc_amd.load()


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:dqmc|detsdw|dethubbard)_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from detqmc_amd._lib import SYMBOLS
    bound = {s[0] for s in SYMBOLS}
    for hdr in ("dqmc_hip.h", "detsdw_host.h", "dethubbard_host.h"):
        names = _declared(hdr)
        assert len(names) > 10
        for nm in names:
            assert hasattr(lib, nm), f"{nm} declared in include/{hdr} but not exported"
            assert nm in bound, f"{nm} declared in include/{hdr} but not bound in detqmc_amd/_lib.py"


def test_rng_restatement_bit_exact_vs_reference(lib):
    z = np.load(os.path.join(GOLDEN, "rng.npz"))
    for key in z.files:
        _, seed, pidx = key.split("_")
        out = np.zeros(len(z[key]))
        assert lib.detsdw_rng_fill(int(seed), int(pidx), out.ctypes.data_as(C.POINTER(C.c_double)), out.size) == 0
        assert np.array_equal(out, z[key]), key


def test_exchange_probability_matches_oracle(lib):
    from detsdw_oracle import replica_exchange_probability
    rng = np.random.default_rng(3)
    for _ in range(50):
        a = rng.normal(size=4) * 3
        assert lib.detsdw_replica_exchange_probability(*a) == replica_exchange_probability(*a)
    assert lib.detsdw_replica_exchange_probability(1.0, 2.0, 1.0, 5.0) == 1.0


def test_fails_loudly_without_gpu_or_with_bad_parameters(lib):
    import torch
    import detqmc_amd
    from detqmc_amd import DqmcError, SDWParams
    # parameter rules of the reference (detsdwparams.cpp:21-140, detmodelparams.h:68-122)
    for bad, frag in [(dict(L=5, beta=2.0), "even linear lattice"),
                      (dict(L=4, beta=2.0, opdim=3, weakZflux=True), "only supported for opdim=2"),
                      (dict(L=4, beta=2.0, m=20), "Only specify one"),
                      (dict(L=4), "either parameter m or beta"),
                      (dict(L=4, beta=2.0, delaySteps=17), "delaySteps"),
                      (dict(L=4, beta=2.0, bc="weird"), "bc")]:
        with pytest.raises(DqmcError) as e:
            detqmc_amd.DetSDW(SDWParams(**bad))
        assert e.value.code == -1 and frag in str(e.value), (bad, str(e.value))
    if not torch.cuda.is_available():
        with pytest.raises(DqmcError) as e:
            detqmc_amd.DetSDW(SDWParams(L=4, beta=2.0))
        assert e.value.code == -5          # DQMC_ENODEV: no silent CPU path
        with pytest.raises(DqmcError):
            detqmc_amd.KernelContext(2, 4, 20, 10, 0.1)
    # execution choices are validated like model parameters (dqmc_tuning, include/dqmc_hip.h)
    with pytest.raises(DqmcError) as e:
        detqmc_amd.KernelContext(2, 4, 20, 10, 0.1, decideThreads=384)
    assert e.value.code == -1 and "decide_threads" in str(e.value)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "detqmc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "detsdw_oracle" not in txt and "dsfmt_oracle" not in txt, f
                assert "oracle/" not in txt.replace("oracle/_ref", ""), f


def test_null_replica_handle_is_rejected(lib):
    import ctypes as C
    from detqmc_amd._lib import detsdw_info
    info = detsdw_info()
    assert lib.detsdw_get_info(None, C.byref(info)) == -1
    assert lib.detsdw_sweep(None, 0) == -1
    assert lib.detsdw_save_state(None, b"/tmp/x") == -1
    assert b"null" in lib.detsdw_last_error()
    assert lib.detsdw_num_chains(None) == 0
