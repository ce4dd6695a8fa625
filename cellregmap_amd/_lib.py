"""ctypes binding of libcrm_hip.so (the C-ABI declared in include/crm_hip.h).

The product path has no CPU fallback: if the library is missing or no MI355X is
visible, calls raise.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcrm_hip.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_long_p = ctypes.POINTER(ctypes.c_long)
vp = ctypes.c_void_p

#: every symbol include/crm_hip.h and include/crm_hip_test.h declare: name -> (restype, argtypes)
SIGNATURES = {
    "crm_last_error": (ctypes.c_char_p, []),
    "crm_version": (ctypes.c_char_p, []),
    "crm_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(vp)]),
    "crm_ctx_destroy": (None, [vp]),
    "crm_ctx_synchronize": (ctypes.c_int, [vp]),
    "crm_ctx_trim": (ctypes.c_int, [vp]),
    "crm_background_create_qs": (ctypes.c_int, [vp, ctypes.c_long, ctypes.c_int, vp, vp, vp, vp,
                                                ctypes.POINTER(vp)]),
    "crm_background_create": (ctypes.c_int, [vp, ctypes.c_long, vp, ctypes.c_int, vp, ctypes.c_long,
                                             ctypes.c_int, vp, ctypes.c_double, ctypes.POINTER(vp)]),
    "crm_background_create_hadamard": (ctypes.c_int, [vp, ctypes.c_long, vp, ctypes.c_int, vp, ctypes.c_int, vp,
                                                      ctypes.c_int, ctypes.c_int, vp, ctypes.c_double,
                                                      ctypes.POINTER(vp)]),
    "crm_background_set_kinship_groups": (ctypes.c_int, [vp, vp, ctypes.c_long, vp, ctypes.c_long, vp, ctypes.c_int]),
    "crm_background_kinship_groups": (ctypes.c_int, [vp]),
    "crm_background_kinship_folded": (ctypes.c_long, [vp]),
    "crm_background_begin": (ctypes.c_int, [vp, ctypes.c_long, vp, ctypes.c_int, vp, ctypes.c_long, vp, ctypes.c_int, vp,
                                            ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_double, ctypes.POINTER(vp)]),
    "crm_background_complete": (ctypes.c_int, [vp, vp]),
    "crm_background_layout": (ctypes.c_int, [vp, c_long_p, c_long_p, c_long_p, c_int_p]),
    "crm_background_export": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp]),
    "crm_background_import": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp]),
    "crm_background_seal": (ctypes.c_int, [vp]),
    "crm_background_destroy": (None, [vp]),
    "crm_background_rank": (ctypes.c_int, [vp, ctypes.c_int]),
    "crm_background_read": (ctypes.c_int, [vp, ctypes.c_int, vp, vp]),
    "crm_gene_create": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.POINTER(vp)]),
    "crm_gene_create_like": (ctypes.c_int, [vp, vp, ctypes.POINTER(vp)]),
    "crm_gene_create_batch": (ctypes.c_int, [vp, vp, ctypes.c_long, ctypes.c_int, ctypes.POINTER(vp)]),
    "crm_gene_destroy": (None, [vp]),
    "crm_panel_create": (ctypes.c_int, [vp, ctypes.c_long, vp, ctypes.c_long, ctypes.c_long,
                                        ctypes.POINTER(vp)]),
    "crm_panel_create_grouped": (ctypes.c_int, [vp, ctypes.c_long, vp, ctypes.c_long, vp, ctypes.c_long,
                                                ctypes.c_long, ctypes.POINTER(vp)]),
    "crm_panel_create_grouped_i8": (ctypes.c_int, [vp, ctypes.c_long, vp, ctypes.c_long, vp, ctypes.c_long, ctypes.c_long,
                                                   ctypes.c_int, ctypes.POINTER(vp)]),
    "crm_panel_create_auto": (ctypes.c_int, [vp, ctypes.c_long, vp, ctypes.c_long, ctypes.c_long, vp, ctypes.c_long,
                                             vp, ctypes.POINTER(vp), c_int_p]),
    "crm_set_donor_collapse": (ctypes.c_int, [vp, ctypes.c_int]),
    "crm_panel_destroy": (None, [vp]),
    "crm_scan_interaction": (ctypes.c_int, [vp, vp, ctypes.c_long, ctypes.c_long] + [vp] * 13),
    "crm_scan_interaction_info": (ctypes.c_int, [vp, vp, ctypes.c_long, ctypes.c_long] + [vp] * 6),
    "crm_scan_interaction_bounds": (ctypes.c_int, [vp, vp, ctypes.c_long, ctypes.c_long] + [vp] * 8),
    "crm_scan_interaction_permuted": (ctypes.c_int, [vp, vp, ctypes.c_long, ctypes.c_long, ctypes.c_int] + [vp] * 8),
    "crm_scan_interaction_multi": (ctypes.c_int, [vp, ctypes.c_int, vp, ctypes.c_long, ctypes.c_long] + [vp] * 8),
    "crm_scan_association": (ctypes.c_int, [vp, vp, ctypes.c_long, ctypes.c_long, ctypes.c_int, vp, vp, vp]),
    "crm_lmm_fit": (ctypes.c_int, [vp, ctypes.c_int, vp, vp]),
    "crm_cov_solve": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, vp, ctypes.c_int, vp]),
    "crm_set_block_variants": (ctypes.c_int, [vp, ctypes.c_int]),
    "crm_set_null_fit_polish": (ctypes.c_int, [vp, ctypes.c_int]),
    "crm_set_progress_callback": (ctypes.c_int, [vp, vp, vp]),
    "crm_set_fast_rotation": (ctypes.c_int, [vp, ctypes.c_int]),
    "crm_kernel_timer_reset": (ctypes.c_int, [vp]),
    "crm_kernel_timer_read": (ctypes.c_int, [vp, c_double_p, c_long_p, c_double_p, c_double_p]),
    "crm_kernel_timer_stop": (ctypes.c_int, [vp]),
    "crm_test_set_form": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    "crm_test_set_contraction": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int]),
    "crm_test_set_shared_h": (ctypes.c_int, [vp, ctypes.c_int]),
    "crm_test_tail_launches": (ctypes.c_long, [vp]),
    "crm_test_spectrum_tail_launches": (ctypes.c_long, [vp]),
    "crm_test_dense_repeats": (ctypes.c_long, [vp]),
    "crm_test_donor_pair_blocks": (ctypes.c_long, [vp]),
    "crm_test_tests_without_pair": (ctypes.c_long, [vp]),
    "crm_test_set_contraction_sync": (ctypes.c_int, [vp, ctypes.c_int]),
    "crm_test_sync_fallbacks": (ctypes.c_long, [vp]),
    "crm_test_overruns": (ctypes.c_long, []),
    "crm_test_set_kinship_route": (ctypes.c_int, [vp, ctypes.c_int]),
    "crm_test_check_context": (ctypes.c_int, [vp]),
    "crm_test_overrun_selftest": (ctypes.c_int, [vp]),
    "crm_test_null_fit_probe": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_double]),
    "crm_test_null_fit_probe_read": (ctypes.c_int, [vp, vp, ctypes.c_long]),
    "crm_test_contract": (ctypes.c_int, [vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, vp, vp, vp,
                                         ctypes.c_int]),
    "crm_test_contract_kr": (ctypes.c_int, [vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            vp, vp, vp, vp]),
    "crm_test_contract_kr_t": (ctypes.c_int, [vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              vp, vp, vp, vp]),
    "crm_test_eigh": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp, vp]),
    "crm_test_eigh2": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp]),
    "crm_test_back_tasks": (ctypes.c_int, [ctypes.c_int, ctypes.c_long, ctypes.c_int, vp, ctypes.c_int, c_int_p]),
    "crm_test_dc_plan": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_int_p, c_double_p, vp, vp, vp,
                                        c_int_p, vp]),
    "crm_test_eigvalsh": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    "crm_test_davies": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp]),
}

_lib = None


class CrmError(RuntimeError):
    pass


def load():
    """Load the shared library and bind every declared symbol (no GPU needed)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CrmError(
            f"{LIB_PATH} is missing: build it with `python -m cellregmap_amd.build` "
            "(there is no CPU fallback for this path)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().crm_last_error()
        raise CrmError(f"libcrm_hip error {rc}: {msg.decode() if msg else ''}")


def f64(a):
    """C-contiguous float64 view/copy."""
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    return None if a is None else a.ctypes.data_as(vp)
