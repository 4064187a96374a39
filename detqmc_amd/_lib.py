"""ctypes binding of the in-tree shared library (include/dqmc_hip.h + include/detsdw_host.h).

There is NO fallback: if libdetqmc_amd.so is missing or cannot be loaded this module raises, and
every compute entry point needs a GPU (dqmc_create returns DQMC_ENODEV without one).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libdetqmc_amd.so")


class DqmcError(RuntimeError):
    """Error code from the C ABI, carrying the library's message (reference: GeneralError)."""

    def __init__(self, code, msg):
        super().__init__(f"[dqmc {code}] {msg}")
        self.code = code


class dqmc_cplx(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class dqmc_tuning(C.Structure):
    _fields_ = [("pipeline", C.c_int32), ("qr_variant", C.c_int32), ("green_variant", C.c_int32),
                ("max_jacobi_sweeps", C.c_int32), ("proposal_budget", C.c_int32), ("decide_threads", C.c_int32), ("reserved", C.c_int32 * 2)]


class dqmc_schedule_info(C.Structure):
    _fields_ = [("pipelined", C.c_int32), ("proposal_budget", C.c_int32), ("blocks_pipelined", C.c_uint64),
                ("blocks_sequential", C.c_uint64), ("qr_block_gram_schmidt", C.c_int32), ("green_lu", C.c_int32),
                ("cholqr_fallbacks", C.c_uint64)]


class dqmc_params(C.Structure):
    _fields_ = [("opdim", C.c_int32), ("L", C.c_int32), ("m", C.c_int32), ("s", C.c_int32),
                ("delaySteps", C.c_int32), ("bc", C.c_int32), ("weakZflux", C.c_int32),
                ("phi2bosons", C.c_int32), ("device", C.c_int32), ("stabilisation", C.c_int32),
                ("cb_none", C.c_int32), ("model", C.c_int32),
                ("dtau", C.c_double), ("r", C.c_double), ("c", C.c_double), ("u", C.c_double),
                ("lambda_", C.c_double),
                ("txhor", C.c_double), ("txver", C.c_double), ("tyhor", C.c_double), ("tyver", C.c_double),
                ("mux", C.c_double), ("muy", C.c_double), ("accRatio", C.c_double), ("cdwU", C.c_double),
                ("rng_window_per_site", C.c_int32), ("reserved_model", C.c_int32),
                ("tuning", dqmc_tuning)]


class dqmc_update_state(C.Structure):
    _fields_ = [("phiDelta", C.c_double), ("targetAccRatio", C.c_double), ("lastAccRatio", C.c_double),
                ("ra_runningAverage", C.c_double), ("ra_values", C.c_double * 100),
                ("ra_samplesAdded", C.c_int32), ("ra_head", C.c_int32),
                ("rng_consumed", C.c_uint64), ("rng_avail", C.c_uint64),
                ("error", C.c_int32), ("reserved", C.c_int32),
                ("angleDelta", C.c_double), ("scaleDelta", C.c_double),
                ("curminAngleDelta", C.c_double), ("curmaxAngleDelta", C.c_double),
                ("curminScaleDelta", C.c_double), ("curmaxScaleDelta", C.c_double),
                ("rot_runningAverage", C.c_double), ("rot_values", C.c_double * 100),
                ("scl_runningAverage", C.c_double), ("scl_values", C.c_double * 100),
                ("rot_samplesAdded", C.c_int32), ("rot_head", C.c_int32), ("scl_samplesAdded", C.c_int32), ("scl_head", C.c_int32)]


class dqmc_profile(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_uint64 * 8),
                ("svd_calls", C.c_uint64), ("svd_sweeps_total", C.c_uint64), ("svd_sweeps_max", C.c_uint64),
                ("qr_calls", C.c_uint64), ("gemm_flops", C.c_double), ("decomp_round_ms", C.c_double),
                ("decomp_rounds", C.c_uint64), ("blocks_nonempty", C.c_uint64), ("chains", C.c_uint64),
                ("updates_accepted", C.c_uint64), ("lu_calls", C.c_uint64),
                ("sub_ms", C.c_double * 4), ("sub_launches", C.c_uint64 * 4), ("sub_flops", C.c_double * 4), ("sub_bytes", C.c_double * 4)]


class detsdw_params(C.Structure):
    _fields_ = [("opdim", C.c_int32), ("L", C.c_int32), ("m", C.c_int32), ("s", C.c_int32),
                ("delaySteps", C.c_int32), ("globalShift", C.c_int32), ("globalUpdateInterval", C.c_int32),
                ("weakZflux", C.c_int32), ("phi2bosons", C.c_int32), ("device", C.c_int32),
                ("simindex", C.c_int32), ("rngSeed", C.c_uint32), ("has_mux_muy", C.c_int32),
                ("updateMethod", C.c_int32), ("bc", C.c_char * 16),
                ("beta", C.c_double), ("dtau", C.c_double),
                ("r", C.c_double), ("c", C.c_double), ("u", C.c_double), ("lambda_", C.c_double),
                ("txhor", C.c_double), ("txver", C.c_double), ("tyhor", C.c_double), ("tyver", C.c_double),
                ("mu", C.c_double), ("mux", C.c_double), ("muy", C.c_double),
                ("accRatio", C.c_double), ("cdwU", C.c_double),
                ("stabilisation", C.c_int32), ("cb_none", C.c_int32),
                ("wolffClusterUpdate", C.c_int32), ("wolffClusterShiftUpdate", C.c_int32),
                ("repeatWolffPerSweep", C.c_int32), ("fermionMeasurements", C.c_int32),
                ("spinProposalMethod", C.c_int32), ("adaptScaleVariance", C.c_int32), ("repeatUpdateInSlice", C.c_int32),
                ("reserved_model", C.c_int32),
                ("tuning", dqmc_tuning)]


class detsdw_info(C.Structure):
    _fields_ = [("opdim", C.c_int32), ("L", C.c_int32), ("N", C.c_int32), ("MSF", C.c_int32),
                ("n_g", C.c_int32), ("m", C.c_int32), ("s", C.c_int32), ("n", C.c_int32),
                ("performedSweeps", C.c_int32), ("lastSweepDir", C.c_int32),
                ("acceptedGlobalShifts", C.c_int32), ("attemptedGlobalShifts", C.c_int32),
                ("currentTimeslice", C.c_int32), ("reserved", C.c_int32),
                ("acceptedWolffClusterUpdates", C.c_int32), ("attemptedWolffClusterUpdates", C.c_int32),
                ("acceptedWolffClusterShiftUpdates", C.c_int32), ("attemptedWolffClusterShiftUpdates", C.c_int32),
                ("addedWolffClusterSize", C.c_double),
                ("beta", C.c_double), ("dtau", C.c_double),
                ("phiDelta", C.c_double), ("lastAccRatioLocal_phi", C.c_double), ("r", C.c_double),
                ("angleDelta", C.c_double), ("scaleDelta", C.c_double),
                ("rngDrawn", C.c_uint64)]


class detsdw_observables(C.Structure):
    _fields_ = [("meanPhi", C.c_double * 3), ("normMeanPhi", C.c_double), ("associatedEnergy", C.c_double),
                ("phiRhoS_Gc", C.c_double), ("phiRhoS_Gs", C.c_double), ("valid", C.c_int32), ("fermionic_valid", C.c_int32),
                ("greenK0", C.c_double), ("greenLocal", C.c_double), ("pairPlusMax", C.c_double), ("pairMinusMax", C.c_double),
                ("occDiffSq", C.c_double)]


class detsdw_control_data(C.Structure):
    _fields_ = [("acceptedGlobalShifts", C.c_int32), ("attemptedGlobalShifts", C.c_int32),
                ("acceptedWolffClusterUpdates", C.c_int32), ("attemptedWolffClusterUpdates", C.c_int32),
                ("acceptedWolffClusterShiftUpdates", C.c_int32), ("attemptedWolffClusterShiftUpdates", C.c_int32),
                ("addedWolffClusterSize", C.c_double),
                ("adjust", dqmc_update_state)]


# every symbol include/*.h declares: (name, restype, argtypes)
_P = C.c_void_p
_DP = C.POINTER(C.c_double)
class dethubbard_params(C.Structure):
    _fields_ = [("L", C.c_int32), ("d", C.c_int32), ("m", C.c_int32), ("s", C.c_int32), ("checkerboard", C.c_int32),
                ("device", C.c_int32), ("simindex", C.c_int32), ("rngSeed", C.c_uint32), ("stabilisation", C.c_int32),
                ("reserved", C.c_int32), ("beta", C.c_double), ("dtau", C.c_double), ("t", C.c_double), ("U", C.c_double),
                ("mu", C.c_double)]


class dethubbard_observables(C.Structure):
    _fields_ = [("occUp", C.c_double), ("occDn", C.c_double), ("occTotal", C.c_double), ("occDouble", C.c_double),
                ("localMoment", C.c_double), ("eKinetic", C.c_double), ("ePotential", C.c_double), ("eTotal", C.c_double),
                ("valid", C.c_int32), ("reserved", C.c_int32)]


class dethubbard_info(C.Structure):
    _fields_ = [("L", C.c_int32), ("N", C.c_int32), ("m", C.c_int32), ("s", C.c_int32), ("n", C.c_int32),
                ("performedSweeps", C.c_int32), ("lastSweepDir", C.c_int32), ("currentTimeslice", C.c_int32),
                ("beta", C.c_double), ("dtau", C.c_double), ("alpha", C.c_double), ("lastAccRatio", C.c_double),
                ("rngDrawn", C.c_uint64)]


SYMBOLS = [
    ("dqmc_create", C.c_int, [C.POINTER(dqmc_params), C.POINTER(_P)]),
    ("dqmc_create_batch", C.c_int, [C.POINTER(dqmc_params), C.c_int, C.POINTER(_P)]),
    ("dqmc_select_chain", C.c_int, [_P, C.c_int]),
    ("dqmc_num_chains", C.c_int, [_P]),
    ("dqmc_destroy", None, [_P]),
    ("dqmc_last_error", C.c_char_p, []),
    ("dqmc_synchronize", C.c_int, [_P]),
    ("dqmc_stream", _P, [_P]),
    ("dqmc_set_fields_host", C.c_int, [_P, _DP]),
    ("dqmc_get_fields_host", C.c_int, [_P, _DP, _DP, _DP]),
    ("dqmc_set_cdwl_host", C.c_int, [_P, _P]),
    ("dqmc_get_cdwl_host", C.c_int, [_P, _P]),
    ("dqmc_set_fields_all_host", C.c_int, [_P, _DP]),
    ("dqmc_get_fields_all_host", C.c_int, [_P, _DP]),
    ("dqmc_bmult_host", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    ("dqmc_udv_decompose_host", C.c_int, [_P, _P, _P, _DP, _P, C.POINTER(C.c_int)]),
    ("dqmc_gemm_host", C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P]),
    ("dqmc_udv_setup", C.c_int, [_P]),
    ("dqmc_advance", C.c_int, [_P, C.c_int, C.c_int]),
    ("dqmc_wrap", C.c_int, [_P, C.c_int, C.c_int]),
    ("dqmc_reset_storage0", C.c_int, [_P]),
    ("dqmc_push_uniforms_host", C.c_int, [_P, _DP, C.c_size_t]),
    ("dqmc_push_uniforms_all_host", C.c_int, [_P, _DP, C.c_size_t]),
    ("dqmc_update_slice", C.c_int, [_P, C.c_int, C.c_int]),
    ("dqmc_update_slice_ex", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("dqmc_get_schedule_info", C.c_int, [_P, C.POINTER(dqmc_schedule_info)]),
    ("dqmc_get_update_states_all_host", C.c_int, [_P, C.POINTER(dqmc_update_state)]),
    ("dqmc_get_update_state_host", C.c_int, [_P, C.POINTER(dqmc_update_state)]),
    ("dqmc_set_update_state_host", C.c_int, [_P, C.POINTER(dqmc_update_state)]),
    ("dqmc_get_green_host", C.c_int, [_P, _P]),
    ("dqmc_set_green_host", C.c_int, [_P, _P, C.c_int]),
    ("dqmc_get_sv_host", C.c_int, [_P, _DP]),
    ("dqmc_get_sv_all_host", C.c_int, [_P, _DP]),
    ("dqmc_get_udv_host", C.c_int, [_P, C.c_int, _P, _DP, _P]),
    ("dqmc_current_timeslice", C.c_int, [_P]),
    ("dqmc_backup", C.c_int, [_P]),
    ("dqmc_restore", C.c_int, [_P]),
    ("dqmc_exchange_action_host", C.c_int, [_P, _DP]),
    ("dqmc_exchange_actions_device", C.c_int, [_P, _P]),
    ("dqmc_phi_action_all_host", C.c_int, [_P, _DP]),
    ("dqmc_shift_fields_all_host", C.c_int, [_P, _DP]),
    ("dqmc_set_exchange_parameter", C.c_int, [_P, C.c_double]),
    ("dqmc_shift_green_symmetric_host", C.c_int, [_P, _P]),
    ("dqmc_measure_reset", C.c_int, [_P]),
    ("dqmc_measure_slice", C.c_int, [_P]),
    ("dqmc_measure_accum_size", C.c_size_t, [_P]),
    ("dqmc_measure_read_host", C.c_int, [_P, _DP]),
    ("dqmc_profile_enable", C.c_int, [_P, C.c_int]),
    ("dqmc_profile_read", C.c_int, [_P, C.POINTER(dqmc_profile)]),
    ("detsdw_create", C.c_int, [C.POINTER(detsdw_params), C.POINTER(_P)]),
    ("detsdw_create_batch", C.c_int, [C.POINTER(detsdw_params), C.c_int, C.POINTER(_P)]),
    ("detsdw_create_batch_ex", C.c_int, [C.POINTER(detsdw_params), C.c_int, C.c_int, C.POINTER(_P)]),
    ("detsdw_num_sub_batches", C.c_int, [_P]),
    ("detsdw_ctx_of_chain", _P, [_P, C.c_int, C.POINTER(C.c_int)]),
    ("detsdw_select_chain", C.c_int, [_P, C.c_int]),
    ("detsdw_num_chains", C.c_int, [_P]),
    ("detsdw_destroy", None, [_P]),
    ("detsdw_last_error", C.c_char_p, []),
    ("detsdw_sweep", C.c_int, [_P, C.c_int]),
    ("detsdw_sweep_thermalization", C.c_int, [_P]),
    ("detsdw_get_info", C.c_int, [_P, C.POINTER(detsdw_info)]),
    ("detsdw_get_observables", C.c_int, [_P, C.POINTER(detsdw_observables)]),
    ("detsdw_get_observable_vector", C.c_int, [_P, C.c_int, _DP]),
    ("detsdw_get_phi", C.c_int, [_P, _DP]),
    ("detsdw_set_phi", C.c_int, [_P, _DP]),
    ("detsdw_get_cdwl", C.c_int, [_P, _P]),
    ("detsdw_set_cdwl", C.c_int, [_P, _P]),
    ("detsdw_get_green", C.c_int, [_P, _P]),
    ("detsdw_get_green_inv_sv", C.c_int, [_P, _DP]),
    ("detsdw_save_configuration_stream_binary", C.c_int, [_P, C.c_char_p]),
    ("detsdw_save_state", C.c_int, [_P, C.c_char_p]),
    ("detsdw_load_state", C.c_int, [_P, C.c_char_p]),
    ("detsdw_rng_rand01", C.c_double, [_P]),
    ("detsdw_ctx", _P, [_P]),
    ("detsdw_get_exchange_parameter_value", C.c_double, [_P]),
    ("detsdw_set_exchange_parameter_value", C.c_int, [_P, C.c_double]),
    ("detsdw_get_exchange_parameter_name", C.c_char_p, [_P]),
    ("detsdw_get_exchange_action_contribution", C.c_int, [_P, _DP]),
    ("detsdw_exchange_actions_device", C.c_int, [_P, _P]),
    ("detsdw_get_control_data", C.c_int, [_P, C.POINTER(detsdw_control_data)]),
    ("detsdw_set_control_data", C.c_int, [_P, C.POINTER(detsdw_control_data)]),
    ("detsdw_replica_exchange_probability", C.c_double, [C.c_double] * 4),
    ("detsdw_rng_fill", C.c_int, [C.c_uint32, C.c_uint32, _DP, C.c_size_t]),
    ("dethubbard_create", C.c_int, [C.POINTER(dethubbard_params), C.c_int, C.POINTER(_P)]),
    ("dethubbard_destroy", None, [_P]),
    ("dethubbard_last_error", C.c_char_p, []),
    ("dethubbard_select_chain", C.c_int, [_P, C.c_int]),
    ("dethubbard_sweep", C.c_int, [_P, C.c_int]),
    ("dethubbard_sweep_thermalization", C.c_int, [_P]),
    ("dethubbard_get_info", C.c_int, [_P, C.POINTER(dethubbard_info)]),
    ("dethubbard_get_auxfield", C.c_int, [_P, _DP]),
    ("dethubbard_get_green", C.c_int, [_P, _DP, _DP]),
    ("dethubbard_get_observables", C.c_int, [_P, C.POINTER(dethubbard_observables)]),
    ("dethubbard_get_zcorr", C.c_int, [_P, _DP]),
    ("dethubbard_save_state", C.c_int, [_P, C.c_char_p]),
    ("dethubbard_load_state", C.c_int, [_P, C.c_char_p]),
    ("dethubbard_rng_rand01", C.c_double, [_P]),
    ("dethubbard_ctx", _P, [_P]),
]

_lib = None


def load():
    """Load the shared library; raises if it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -m detqmc_amd.build` "
                          "(there is no CPU fallback for the DQMC kernels)")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, host=False):
    if rc != 0:
        lib = load()
        msg = (lib.dethubbard_last_error() if host == "hubbard" else lib.detsdw_last_error() if host else lib.dqmc_last_error()) or b""
        raise DqmcError(rc, msg.decode("utf-8", "replace"))
