"""
ctypes binding of librevrand_hip.so (include/revrand_hip.h) -- the only door between the
Python classes of this package and the GPU.  No torch, no numpy-side fallback: if the
library or a gfx950 device is missing, the calls raise.

A device context is created lazily, once per *process* (sklearn may fork workers, SURVEY
8b), and is never pickled.
"""
import contextlib
import ctypes
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librevrand_hip.so")

RR_F32, RR_F64 = 0, 1
RR_F32P64 = 2  # rr_rff_create's compute only: f32 pipeline, float64 phases (include/revrand_hip.h)
_NP2RR = {np.dtype(np.float32): RR_F32, np.dtype(np.float64): RR_F64}
_RR2NP = {RR_F32: np.float32, RR_F64: np.float64}

_c_void_pp = ctypes.POINTER(ctypes.c_void_p)
_c_double_p = ctypes.POINTER(ctypes.c_double)

# name -> (restype, argtypes); every symbol include/revrand_hip.h declares
SIGNATURES = {
    "rr_abi_version": (ctypes.c_int, []),
    "rr_build_flags": (ctypes.c_int, []),
    "rr_debug_launch_checks": (ctypes.c_int64, []),
    "rr_legacy_permutation": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rr_legacy_randn": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int, ctypes.c_int64, ctypes.c_int]),
    "rr_last_error": (ctypes.c_char_p, []),
    "rr_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "rr_ctx_create": (ctypes.c_int, [ctypes.c_int, _c_void_pp]),
    "rr_ctx_destroy": (None, [ctypes.c_void_p]),
    "rr_ctx_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "rr_ctx_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int),
                                   ctypes.POINTER(ctypes.c_uint64)]),
    "rr_ctx_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
    "rr_ctx_pci_bus_id": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]),
    "rr_peer_access": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "rr_malloc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, _c_void_pp]),
    "rr_free": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rr_memset": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]),
    "rr_memcpy_h2d": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "rr_memcpy_d2h": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "rr_timer_start": (ctypes.c_int, [ctypes.c_void_p]),
    "rr_timer_stop": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "rr_rff_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p, _c_void_pp]),
    "rr_basis_destroy": (None, [ctypes.c_void_p]),
    "rr_rff_padded_dim": (ctypes.c_int, [ctypes.c_void_p]),
    "rr_upload_matrix": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _c_void_pp]),
    "rr_upload_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]),
    "rr_rff_transform": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_int64]),
    "rr_rff_grad": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                   ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                   ctypes.c_int]),
    "rr_rff_transform_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                            ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_int64]),
    "rr_rff_gram_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_rff_gram_timings": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float),
                                           ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                           ctypes.POINTER(ctypes.c_int)]),
    "rr_symmetrize_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    "rr_rff_gram": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_dense_gram": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                     ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p]),
    "rr_dense_predict": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, _c_void_pp]),
    "rr_featmat_destroy": (None, [ctypes.c_void_p]),
    "rr_featmat_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    "rr_featmat_put_rff": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]),
    "rr_featmat_put_linear": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "rr_featmat_put_fastfood": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]),
    "rr_featmat_put_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                           ctypes.c_int64, ctypes.c_int64]),
    "rr_featmat_gram": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat64_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, _c_void_pp]),
    "rr_featmat64_destroy": (None, [ctypes.c_void_p]),
    "rr_featmat64_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    "rr_featmat64_put_rff": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]),
    "rr_featmat64_put_linear": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "rr_featmat64_put_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_int64]),
    "rr_featmat64_gram": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat64_pass2_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "rr_featmat64_pass2_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "rr_featmat64_pass2_rff": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "rr_featmat64_pass2_end": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat64_predict_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_rff_elbo_pass2_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat_pass2_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat_pass2_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "rr_featmat_pass2_rff": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "rr_featmat_pass2_plan_rff": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                 ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "rr_featmat_pass2_rows_planned": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "rr_featmat_pass2_end": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat_predict_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat_glm_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat_glm_step_sampled": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat_glm_step_draws": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat_glm_step_draws_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_gather_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                      ctypes.c_void_p]),
    "rr_featmat_glm_rff": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "rr_featmat_glm_plan_rff": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "rr_featmat_glm_edphi": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "rr_featmat_project": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "rr_glm_sgd_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]),
    "rr_glm_sgd_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                       ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]),
    "rr_glm_sgd_dist_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                            ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]),
    "rr_glm_sgd_group_step": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64]),
    "rr_glm_sgd_read": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.POINTER(ctypes.c_int64)]),
    "rr_glm_sgd_objective": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]),
    "rr_glm_sgd_destroy": (None, [ctypes.c_void_p]),
    "rr_glm_svi_supported": (ctypes.c_int, [ctypes.c_int] * 7),
    "rr_glm_svi_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_int64, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p)]),
    "rr_glm_svi_set_start": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_glm_svi_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                      ctypes.c_uint64]),
    "rr_glm_svi_starts": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]),
    "rr_glm_svi_read": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.POINTER(ctypes.c_int64)]),
    "rr_glm_svi_destroy": (None, [ctypes.c_void_p]),
    "rr_posterior_available": (ctypes.c_int, []),
    "rr_set_gram_engine": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rr_get_gram_engine": (ctypes.c_int, [ctypes.c_void_p]),
    "rr_set_deterministic": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rr_get_deterministic": (ctypes.c_int, [ctypes.c_void_p]),
    "rr_posterior_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p]),
    "rr_rff_elbo_pass2_devc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat_pass2_begin_devc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat_predict_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "rr_rff_grad_contract": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                            ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]),
    "rr_rff_predict_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_rff_predict_devc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_variance_factor_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.POINTER(ctypes.c_int)]),
    "rr_rff_predict_devb": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                           ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_void_p]),
    "rr_featmat_predict_begin_b": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "rr_rff_predict_mean_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                               ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "rr_fastfood_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          _c_void_pp]),
    "rr_fastfood_transform": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.c_int, ctypes.c_int64]),
    "rr_fastfood_transform_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.c_int, ctypes.c_int64]),
    "rr_fastfood_vx": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_int, ctypes.c_int64]),
    "rr_gm_transform": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_int64]),
    "rr_gm_grad": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_int]),
    "rr_fastfood_gm_transform": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int64]),
    "rr_fastfood_gm_transform_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                    ctypes.c_int64]),
    "rr_featmat_put_fastfood_gm": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]),
    "rr_hadamard": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                   ctypes.c_int, ctypes.c_void_p]),
    "rr_rff_gram_kernel_name": (ctypes.c_char_p, [ctypes.c_void_p]),
    # multi-GPU exchange (RCCL bound directly; revrand_amd/parallel.py)
    "rr_comm_load": (ctypes.c_int, [ctypes.c_char_p]),
    "rr_comm_version": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_size_t]),
    "rr_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "rr_comm_init_rank": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, _c_void_pp]),
    "rr_comm_destroy": (None, [ctypes.c_void_p]),
    "rr_comm_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "rr_comm_allreduce_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]),
    "rr_comm_allreduce_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]),
    "rr_comm_broadcast_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]),
    "rr_comm_barrier": (ctypes.c_int, [ctypes.c_void_p]),
    "rr_stats_msg_count": (ctypes.c_int64, [ctypes.c_int64]),
    "rr_stats_pack_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]),
    "rr_stats_unpack_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p]),
    "rr_comm_reduce_stats_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    # one process, several GPUs (revrand_amd/multigpu.py)
    "rr_comm_init_all": (ctypes.c_int, [ctypes.c_int, _c_void_pp, ctypes.c_int, _c_void_pp]),
    "rr_comm_transport": (ctypes.c_int, [ctypes.c_void_p]),
    "rr_comm_group_allreduce_dev": (ctypes.c_int, [_c_void_pp, ctypes.c_int, _c_void_pp, ctypes.c_int64, ctypes.c_int]),
    "rr_comm_group_broadcast_dev": (ctypes.c_int, [_c_void_pp, ctypes.c_int, _c_void_pp, ctypes.c_int64, ctypes.c_int]),
    "rr_comm_group_reduce_stats_dev": (ctypes.c_int, [_c_void_pp, ctypes.c_int, ctypes.c_int64, _c_void_pp, _c_void_pp,
                                                      _c_void_pp, _c_double_p, _c_void_pp, _c_double_p]),
}


class HipError(RuntimeError):
    """A call into librevrand_hip.so failed; the message is rr_last_error()."""

    def __init__(self, code, message):
        super().__init__("librevrand_hip: %s (status %d)" % (message, code))
        self.code = code


_lib = None
_lib_lock = threading.Lock()


def _select_hip_runtime():
    """One HIP runtime per process, and by default the one librevrand_hip.so was BUILT against (its RUNPATH:
    /opt/rocm).  PyTorch-ROCm wheels bundle their own libamdhip64.so with the same soname: if torch is already imported
    the loader hands that copy to our library too (nothing to do, and nothing we could do); if torch will be imported
    LATER in this process, set ``RR_HIP_RUNTIME=torch`` (or import torch first) -- otherwise torch would bring a second
    runtime + HSA instance into the process and see no GPU.  ``RR_HIP_RUNTIME=torch`` loads torch's bundled copy first
    (no ``import torch``); ``RR_HIP_RUNTIME=system`` (the default) changes nothing."""
    mode = os.environ.get("RR_HIP_RUNTIME", "system")
    if mode not in ("system", "torch", ""):
        raise ImportError("RR_HIP_RUNTIME must be 'system' or 'torch', got %r" % mode)
    if mode != "torch" or "torch" in sys.modules:
        return
    cand = _torch_lib("libamdhip64.so")
    if cand is None:
        raise ImportError("RR_HIP_RUNTIME=torch, but no torch wheel with a bundled libamdhip64.so is installed")
    ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def _mapped(key):
    """Paths of the shared objects mapped into this process whose file name contains `key`."""
    try:
        with open("/proc/self/maps") as f:
            return sorted({line.split()[-1] for line in f if key in line and "/" in line.split()[-1]})
    except OSError:
        return []


def hip_runtime_path():
    """The libamdhip64 this process runs on (after load_library()); several paths = a broken process, all are reported."""
    paths = _mapped("libamdhip64")
    return paths[0] if len(paths) == 1 else (paths or None)


_bound_runtime = None   # libamdhip64 path(s) mapped at the moment librevrand_hip.so was loaded


def _runtime_is_torchs():
    """Does librevrand_hip.so run on the torch wheel's HIP runtime?  Decided by what was mapped when the library was LOADED:
    a process that loads it first (on /opt/rocm's runtime) and imports torch afterwards has both copies mapped, and RCCL
    must still be the one paired with OURS -- the wheel's librccl on /opt/rocm's runtime sees no device
    (ncclCommInitAll: 'no ROCm-capable device is detected'; found by the full GPU suite, where a torch-launcher test ran
    before the in-process RCCL group's)."""
    cand = _torch_lib("libamdhip64.so")
    if cand is None:
        return False
    tdir = os.path.realpath(os.path.dirname(cand))
    paths = _bound_runtime if _bound_runtime else _mapped("libamdhip64")
    return any(os.path.realpath(os.path.dirname(p)) == tdir for p in paths)


def rccl_library_path():
    """The librccl that belongs to the HIP runtime this process runs on: $RR_RCCL_LIB; the copy bundled with the torch
    wheel when (and only when) the process runs on that wheel's HIP runtime; None = the library's own search (an already
    loaded librccl.so.1, then /opt/rocm/lib)."""
    if os.environ.get("RR_RCCL_LIB"):
        return os.environ["RR_RCCL_LIB"]
    if _runtime_is_torchs():
        return _torch_lib("librccl.so")
    # the copy next to OUR runtime, by path: a bare dlopen("librccl.so.1") hands back whichever copy is already in the
    # process -- the torch wheel's, once anybody imported torch
    for rt in (_bound_runtime or _mapped("libamdhip64")):
        cand = os.path.join(os.path.dirname(os.path.realpath(rt)), "librccl.so.1")
        if os.path.exists(cand):
            return cand
    return None


def _torch_lib(name):
    """Path of a library bundled with an installed torch wheel, or None."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return None
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
        return cand if os.path.exists(cand) else None
    except (ImportError, ValueError):
        return None


RR_ERR_NOT_POSDEF = -6


def posterior_available(F=None):
    """Should the posterior of an F x F system be formed on the device (rr_posterior_dev)?

    ``RR_POSDEF=host`` never, ``RR_POSDEF=device`` always; by default from F >= 256 (``RR_POSDEF_MIN_F``): below
    that the host LAPACK solve takes well under a millisecond and the device pipeline is launch-bound."""
    mode = os.environ.get("RR_POSDEF", "")
    if mode == "host":
        return False
    if mode != "device" and F is not None and F < int(os.environ.get("RR_POSDEF_MIN_F", "256")):
        return False
    return bool(load_library().rr_posterior_available())


def load_library(path=None):
    """Load the shared library and declare every prototype.  Raises if it is not built."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        _select_hip_runtime()
        p = path or os.environ.get("REVRAND_HIP_LIB", LIB_PATH)
        if not os.path.exists(p):
            raise ImportError(
                "librevrand_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C revrand_amd/csrc`; there is no CPU fallback for this path" % p)
        lib = ctypes.CDLL(p)
        global _bound_runtime
        if _bound_runtime is None:   # the HIP runtime OUR library is bound to: what is mapped now (a torch imported later
            _bound_runtime = _mapped("libamdhip64")   # maps its own copy next to it, which is not ours)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if lib.rr_abi_version() != 1:
            raise ImportError("librevrand_hip.so ABI version %d != 1" % lib.rr_abi_version())
        if path is None:
            _lib = lib
        return lib


def _check(lib, rc):
    if rc != 0:
        msg = lib.rr_last_error()
        raise HipError(rc, msg.decode("utf-8", "replace") if msg else "unknown error")


def rr_dtype(np_dtype):
    try:
        return _NP2RR[np.dtype(np_dtype)]
    except KeyError:
        raise TypeError("only float32/float64 arrays can cross the HIP boundary, got %s" % np_dtype)


def as_float_matrix(X):
    """X as a float32/float64 2-d array whose rows are contiguous (other dtypes -> float64).

    Row-sliced views with a larger row stride are passed through with their leading dimension.
    """
    X = np.asarray(X)
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64)
    if X.ndim != 2:
        raise ValueError("expected a 2-d array")
    N, d = X.shape
    cols_ok = d == 0 or X.strides[1] == X.itemsize
    rows_ok = N <= 1 or (X.strides[0] % X.itemsize == 0 and X.strides[0] >= d * X.itemsize)
    if not (cols_ok and rows_ok):
        X = np.ascontiguousarray(X)
    return X


def _ld(X):
    return X.strides[0] // X.itemsize if X.shape[0] > 1 else max(X.shape[1], 1)


class DeviceBuffer(object):
    """A device allocation owned by Python (freed on garbage collection)."""

    def __init__(self, dev, ptr, nbytes):
        self.dev, self.ptr, self.nbytes = dev, ptr, nbytes

    def free(self):
        if self.ptr:
            try:
                self.dev.lib.rr_free(self.dev.ctx, self.ptr)
            except Exception:
                pass
            self.ptr = None

    def __del__(self):
        self.free()


class DeviceCovariance(DeviceBuffer):
    """A posterior covariance (F, F) float64 resident in HBM, with -- on first use -- its prediction factor
    (rr_variance_factor_dev): C = M M^T, so that predict_moments' variance is the sum of squares ||phi^T M||^2 instead of
    the float32 quadratic form phi^T C phi, whose error scales with |phi|^T |C| |phi| (cancellation for badly scaled C)."""

    def __init__(self, dev, C):
        C = np.ascontiguousarray(C, dtype=np.float64)
        if C.ndim != 2 or C.shape[0] != C.shape[1]:
            raise ValueError("a covariance is a square matrix")
        buf = dev.upload_vector(C.ravel())
        super().__init__(dev, buf.ptr, buf.nbytes)
        buf.ptr = None  # ownership moved here
        self.F = int(C.shape[0])
        self.dtype = np.dtype(np.float64)
        self._factor = None

    def factor(self):
        """(device float32 (Fp, Fp) upper-triangular factor, form): form 1 -> Vf = rowsum((Phi B)^2); form 0 (C not safely
        positive definite) -> B is the triangular form of C and Vf = rowsum((Phi B) o Phi)."""
        if self._factor is None:
            Fp = (self.F + 255) // 256 * 256
            B = self.dev.malloc(Fp * Fp * 4)
            form = ctypes.c_int()
            _check(self.dev.lib, self.dev.lib.rr_variance_factor_dev(self.dev.ctx, self.F, self.ptr, B.ptr, ctypes.byref(form)))
            self._factor = (B, form.value)
        return self._factor

    def free(self):
        f, self._factor = getattr(self, "_factor", None), None
        if f is not None:
            f[0].free()
        super().free()


class DeviceMatrix(DeviceBuffer):
    """Row-major (N, d) matrix on the device with leading dimension `ld` (pad columns zero)."""

    def __init__(self, dev, ptr, shape, ld, dtype):
        super().__init__(dev, ptr, shape[0] * ld * np.dtype(dtype).itemsize)
        self.shape, self.ld, self.dtype = tuple(shape), ld, np.dtype(dtype)


class Device(object):
    """One rr_ctx: a GPU, a stream, and helpers.  Use get_device()."""

    _uids = iter(range(1, 1 << 62))

    def __init__(self, index):
        self.lib = load_library()
        self.index = index
        self.pid = os.getpid()
        self.uid = ("ctx", next(Device._uids))  # distinguishes two contexts on one GPU (device_key)
        ctx = ctypes.c_void_p()
        _check(self.lib, self.lib.rr_ctx_create(index, ctypes.byref(ctx)))
        self.ctx = ctx
        name = ctypes.create_string_buffer(64)
        cus = ctypes.c_int()
        hbm = ctypes.c_uint64()
        _check(self.lib, self.lib.rr_ctx_info(ctx, name, ctypes.byref(cus), ctypes.byref(hbm)))
        self.name, self.compute_units, self.hbm_bytes = name.value.decode(), cus.value, hbm.value

    # -- where the GPU sits (bench.py's preflight, multi-GPU placement) ------------------
    @property
    def pci_bus_id(self):
        buf = ctypes.create_string_buffer(32)
        _check(self.lib, self.lib.rr_ctx_pci_bus_id(self.ctx, buf, None))
        return buf.value.decode()

    @property
    def numa_node(self):
        """Host memory node next to this GPU (None when the platform does not say)."""
        try:
            with open("/sys/bus/pci/devices/%s/numa_node" % self.pci_bus_id.lower()) as f:
                node = int(f.read().strip())
            return node if node >= 0 else None
        except (OSError, ValueError):
            return None

    def numa_cpus(self):
        """The CPUs of that node (a set), or None."""
        node = self.numa_node
        if node is None:
            return None
        try:
            with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
                text = f.read().strip()
        except OSError:
            return None
        cpus = set()
        for part in text.split(","):
            if part:
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        return cpus or None

    def can_access_peer(self, other_index):
        can = ctypes.c_int()
        _check(self.lib, self.lib.rr_peer_access(self.index, int(other_index), ctypes.byref(can)))
        return bool(can.value)

    # -- arithmetic of the f32 Gram (include/revrand_hip.h: RR_GRAM_*) -------------
    GRAM_ENGINES = {"f32": 0, "bf16x3": 3, "bf16x4": 4, "fp16x3": 5}

    @property
    def gram_engine(self):
        code = self.lib.rr_get_gram_engine(self.ctx)
        return {v: k for k, v in self.GRAM_ENGINES.items()}[code]

    def set_gram_engine(self, name):
        """'f32' (default; f32 MFMA), 'bf16x3' or 'bf16x4' (split-bf16 products on the bf16 matrix pipe, f32
        accumulation; ~4e-6 / ~2e-6 of max|G| away from the f32 engine).  Returns the previous engine."""
        if name not in self.GRAM_ENGINES:
            raise ValueError("gram engine must be one of %s" % sorted(self.GRAM_ENGINES))
        prev = self.gram_engine
        _check(self.lib, self.lib.rr_set_gram_engine(self.ctx, self.GRAM_ENGINES[name]))
        return prev

    def close(self):
        """Destroy the context (stream, events, scratch).  Buffers allocated through it must have been freed."""
        ctx, self.ctx = self.ctx, None
        if ctx is not None and self.pid == os.getpid():
            self.lib.rr_ctx_destroy(ctx)

    # -- run-to-run reproducibility (include/revrand_hip.h: rr_set_deterministic) ------------
    @property
    def deterministic(self):
        return bool(self.lib.rr_get_deterministic(self.ctx))

    def set_deterministic(self, on=True):
        """Ordered partial sums instead of floating-point atomics in the Gram, posterior and second-pass kernels: the same
        bits every run (and on every rank holding the same statistics).  Also RR_DETERMINISTIC=1.  Returns the previous
        setting."""
        prev = self.deterministic
        _check(self.lib, self.lib.rr_set_deterministic(self.ctx, 1 if on else 0))
        return prev

    # -- memory ---------------------------------------------------------------
    def malloc(self, nbytes):
        p = ctypes.c_void_p()
        _check(self.lib, self.lib.rr_malloc(self.ctx, nbytes, ctypes.byref(p)))
        return DeviceBuffer(self, p, nbytes)

    def zeros(self, nbytes):
        buf = self.malloc(nbytes)
        _check(self.lib, self.lib.rr_memset(self.ctx, buf.ptr, 0, nbytes))
        return buf

    def memset(self, buf, nbytes=None):
        _check(self.lib, self.lib.rr_memset(self.ctx, buf.ptr, 0, buf.nbytes if nbytes is None else nbytes))

    def upload_matrix(self, X, ld_dev=None):
        X = as_float_matrix(X)
        N, d = X.shape
        ld_dev = max(d, 1) if ld_dev is None else ld_dev
        p = ctypes.c_void_p()
        _check(self.lib, self.lib.rr_upload_matrix(self.ctx, X.ctypes.data_as(ctypes.c_void_p), rr_dtype(X.dtype),
                                                   N, max(d, 1), _ld(X), ld_dev, ctypes.byref(p)))
        return DeviceMatrix(self, p, (N, d), ld_dev, X.dtype)

    def empty_matrix(self, N, d, dtype, ld_dev=None):
        """Zero-initialised device (N, d) matrix to be filled with upload_rows."""
        ld_dev = max(d, 1) if ld_dev is None else ld_dev
        nbytes = max(N * ld_dev, 1) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        _check(self.lib, self.lib.rr_malloc(self.ctx, nbytes, ctypes.byref(p)))
        _check(self.lib, self.lib.rr_memset(self.ctx, p, 0, nbytes))
        return DeviceMatrix(self, p, (N, d), ld_dev, dtype)

    def upload_rows(self, dmat, row0, X):
        X = as_float_matrix(X)
        if X.dtype != dmat.dtype or X.shape[1] != dmat.shape[1] or row0 + X.shape[0] > dmat.shape[0]:
            raise ValueError("upload_rows: dtype/shape mismatch")
        _check(self.lib, self.lib.rr_upload_rows(self.ctx, dmat.ptr, dmat.ld, row0,
                                                 X.ctypes.data_as(ctypes.c_void_p), rr_dtype(X.dtype),
                                                 X.shape[0], X.shape[1], _ld(X)))

    def upload_vector(self, y, dtype=None):
        y = np.ascontiguousarray(y, dtype=dtype)
        buf = self.malloc(max(y.nbytes, 1))
        if y.nbytes:
            _check(self.lib, self.lib.rr_memcpy_h2d(self.ctx, buf.ptr, y.ctypes.data_as(ctypes.c_void_p), y.nbytes))
        buf.shape, buf.dtype = y.shape, y.dtype
        return buf

    def download(self, buf, shape, dtype, offset_bytes=0):
        out = np.empty(shape, dtype=dtype)
        if out.nbytes:
            src = ctypes.c_void_p(buf.ptr.value + offset_bytes)
            _check(self.lib, self.lib.rr_memcpy_d2h(self.ctx, out.ctypes.data_as(ctypes.c_void_p), src, out.nbytes))
        return out

    def sync(self):
        _check(self.lib, self.lib.rr_ctx_sync(self.ctx))

    def gather_rows(self, src, didx, rows, dst):
        """dst[r] = src[didx[r]] for two DeviceMatrix objects of the same dtype and leading dimension (async)."""
        if src.ld != dst.ld or src.dtype != dst.dtype or src.dtype.itemsize not in (4, 8):
            raise ValueError("gather_rows: float32 / float64 matrices of one dtype with equal leading dimensions expected")
        _check(self.lib, self.lib.rr_gather_rows(self.ctx, src.ptr, _ptr(didx), rows, src.ld * (src.dtype.itemsize // 4), dst.ptr))

    def posterior(self, F, dG, db, iL, var, dC):
        """rr_posterior_dev: (m, diagC, log|iC|, sum(G o C)) with C left in the device buffer dC, or None when the
        matrix is not safely positive definite (the caller then takes the host SVD route)."""
        iL = np.ascontiguousarray(iL, dtype=np.float64)
        m, dg, scal = np.empty(F), np.empty(F), np.zeros(3)
        rc = self.lib.rr_posterior_dev(self.ctx, F, _ptr(dG), _ptr(db), iL.ctypes.data_as(ctypes.c_void_p), float(var),
                                       _ptr(dC), m.ctypes.data_as(ctypes.c_void_p), dg.ctypes.data_as(ctypes.c_void_p),
                                       scal.ctypes.data_as(ctypes.c_void_p))
        if rc == RR_ERR_NOT_POSDEF:
            return None
        _check(self.lib, rc)
        return m, dg, float(scal[0]), float(scal[1])

    def timer_start(self):
        _check(self.lib, self.lib.rr_timer_start(self.ctx))

    def timer_stop(self):
        ms = ctypes.c_float()
        _check(self.lib, self.lib.rr_timer_stop(self.ctx, ctypes.byref(ms)))
        return ms.value


_devices = {}
_tls = threading.local()  # .dev: the Device every `device=None` lookup of THIS thread resolves to (device_scope)


@contextlib.contextmanager
def device_scope(dev):
    """Make `dev` (a Device) the default device of the calling thread for a block: every handle, buffer and fit state
    created inside with `device=None` lives on it.  How the members of an in-process device group (multigpu.DeviceGroup:
    one host thread per member) run the single-device classes unchanged."""
    prev = getattr(_tls, "dev", None)
    _tls.dev = dev
    try:
        yield dev
    finally:
        _tls.dev = prev


def device_key():
    """(pid, context id) of the calling thread's default device: the key of per-process, per-context caches (a basis'
    device handles), so that a basis used on several members of a device group holds one handle per member."""
    dev = getattr(_tls, "dev", None)
    return (os.getpid(), dev.uid if dev is not None else ("default", default_device_index()))


def default_device_index():
    for var in ("REVRAND_HIP_DEVICE", "LOCAL_RANK"):
        if os.environ.get(var, "") != "":
            return int(os.environ[var])
    return 0


def get_device(index=None):
    """The process-local Device for GPU `index` (default: $REVRAND_HIP_DEVICE, $LOCAL_RANK, 0)."""
    if isinstance(index, Device):
        return index
    if index is None:
        cur = getattr(_tls, "dev", None)
        if cur is not None and cur.pid == os.getpid():
            return cur
    index = default_device_index() if index is None else int(index)
    key = (os.getpid(), index)
    dev = _devices.get(key)
    if dev is None:
        dev = Device(index)
        _devices[key] = dev
    return dev


_upload_devices = {}


def get_upload_device(index=None):
    """A SECOND process-local context (own stream) on GPU `index`, for host threads that copy data up while the main
    context's stream runs kernels (the GLM's draw uploads).  Created once per process and device, like `get_device`."""
    index = default_device_index() if index is None else int(index)
    key = (os.getpid(), index)
    dev = _upload_devices.get(key)
    if dev is None:
        dev = _upload_devices[key] = Device(index)
    return dev


def legacy_randn(random_state, n, dtype=np.float32, threads=2, out=None):
    """`random_state.randn(n)` (a NumPy legacy RandomState), bit for bit and leaving the same state behind, through
    rr_legacy_randn: the MT19937 words, the polar method's candidate pairs and their accept count on this thread (array loops,
    AVX2 where the host has it), the pick of the accepted pairs with their square roots and logarithms on `threads` worker
    threads (two keep up on the GPU box's host: 2.2 ms median per 1 024 000 values against NumPy's 9.3; rounds 2-3, with the
    accept / reject walk on this thread: 4.4).  Anything but a plain MT19937 RandomState falls back to NumPy itself (same
    values either way).  out: a contiguous array of n values of `dtype` to fill instead of a new one (a caller drawing 4 MB per SVI
    step keeps a ring of them: fresh arrays of that size are mapped and unmapped by the allocator, and in a process with
    hundreds of threads the page faults and TLB shootdowns of that stall every thread for tens of milliseconds at a time)."""
    dtype = np.dtype(dtype)
    try:
        st = random_state.get_state(legacy=True)
    except TypeError:
        st = random_state.get_state()
    if out is not None and not (isinstance(out, np.ndarray) and out.dtype == dtype and out.size == n and out.flags.c_contiguous):
        raise ValueError("out must be a contiguous array of n values of the requested dtype")
    if n <= 0 or not isinstance(st, tuple) or st[0] != "MT19937" or dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        vals = random_state.randn(n).astype(dtype, copy=False)
        if out is None:
            return vals
        out.reshape(-1)[:] = vals
        return out
    lib = load_library()
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos, has, g = ctypes.c_int32(int(st[2])), ctypes.c_int32(int(st[3])), ctypes.c_double(float(st[4]))
    out = np.empty(n, dtype=dtype) if out is None else out
    threads = int(os.environ.get("RR_RANDN_THREADS", threads))  # (measurement switch)
    _check(lib, lib.rr_legacy_randn(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), ctypes.byref(has), ctypes.byref(g),
                                    out.ctypes.data_as(ctypes.c_void_p), rr_dtype(dtype), n, threads))
    random_state.set_state(("MT19937", key, pos.value, has.value, g.value))
    return out


def legacy_permutation(random_state, n, out=None):
    """`random_state.permutation(n)` for an integer n (a NumPy legacy RandomState), bit for bit and leaving the same state
    behind, through rr_legacy_permutation -- WITHOUT the GIL, which NumPy's shuffle loop holds throughout (18 ms at n = 2M: the
    epoch boundary of config 5's minibatch stream, where every thread of the fit stood still).  Anything but a plain MT19937
    RandomState, and short permutations, go to NumPy itself (same values either way).  out: an int64 array of n entries to fill
    (see legacy_randn)."""
    n = int(n)
    if out is not None and not (isinstance(out, np.ndarray) and out.dtype == np.int64 and out.shape == (n,) and out.flags.c_contiguous):
        raise ValueError("out must be a contiguous int64 array of n entries")
    try:
        st = random_state.get_state(legacy=True)
    except TypeError:
        st = random_state.get_state()
    if n < 4096 or n > (1 << 32) or not isinstance(st, tuple) or st[0] != "MT19937":
        vals = random_state.permutation(n)
        if out is None:
            return vals
        out[:] = vals
        return out
    lib = load_library()
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = ctypes.c_int32(int(st[2]))
    out = np.empty(n, dtype=np.int64) if out is None else out
    _check(lib, lib.rr_legacy_permutation(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), n, out.ctypes.data_as(ctypes.c_void_p)))
    random_state.set_state(("MT19937", key, pos.value, int(st[3]), float(st[4])))
    return out


@contextlib.contextmanager
def gram_engine_scope(name, index=None):
    """Select the arithmetic of the f32 Gram / U = Phi C / GLM GEMMs (Device.set_gram_engine) for a block; None = leave
    the context's setting (RR_SYRK_ENGINE or an earlier set_gram_engine) alone.  index: a GPU index, a Device, or a list
    of Devices (the members of a device group)."""
    if name is None:
        yield
        return
    devs = list(index) if isinstance(index, (list, tuple)) else [get_device(index)]
    prev = [dev.set_gram_engine(name) for dev in devs]
    try:
        yield
    finally:
        for dev, p in zip(devs, prev):
            dev.set_gram_engine(p)


def device_available():
    """True iff the library loads and sees at least one HIP device (never raises)."""
    try:
        lib = load_library()
        n = ctypes.c_int()
        return lib.rr_device_count(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


class DeviceView(object):
    """Non-owning window of device memory: rows [r0, r0 + rows) of a DeviceMatrix, or elements of a vector."""

    def __init__(self, base, r0, rows):
        self.base = base  # keeps the owner alive
        if isinstance(base, DeviceMatrix):
            self.ld, self.shape = base.ld, (rows, base.shape[1])
            off = r0 * base.ld
        else:
            self.shape = (rows,)
            off = r0
        self.dtype = np.dtype(base.dtype)
        self.ptr = ctypes.c_void_p(base.ptr.value + off * self.dtype.itemsize)


def _ptr(x):
    """DeviceBuffer / DeviceView / ctypes pointer / integer address / None -> c_void_p (or None)."""
    if x is None:
        return None
    if isinstance(x, (DeviceBuffer, DeviceView)):
        return x.ptr
    if isinstance(x, ctypes.c_void_p):
        return x
    return ctypes.c_void_p(int(x))


def _lenscale_arg(lenscale):
    ls = np.ascontiguousarray(np.atleast_1d(np.asarray(lenscale, dtype=np.float64)))
    return ls, ls.ctypes.data_as(ctypes.c_void_p), int(ls.size)


def dense_gram(Phi, y=None, device=None):
    """(Phi^T Phi, Phi^T y, y^T y) of an arbitrary host feature matrix on the GPU (rr_dense_gram)."""
    dev = get_device(device)
    Phi = as_float_matrix(Phi)
    N, F = Phi.shape
    G = np.empty((F, F))
    if y is None:
        _check(dev.lib, dev.lib.rr_dense_gram(dev.ctx, Phi.ctypes.data_as(ctypes.c_void_p), rr_dtype(Phi.dtype), N, F,
                                              _ld(Phi), None, G.ctypes.data_as(ctypes.c_void_p), None, None))
        return G, None, None
    y = np.ascontiguousarray(y, dtype=Phi.dtype).ravel()
    if y.shape[0] != N:
        raise ValueError("Phi and y have inconsistent numbers of rows")
    b, yty = np.empty(F), np.empty(1)
    _check(dev.lib, dev.lib.rr_dense_gram(dev.ctx, Phi.ctypes.data_as(ctypes.c_void_p), rr_dtype(Phi.dtype), N, F,
                                          _ld(Phi), y.ctypes.data_as(ctypes.c_void_p),
                                          G.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                                          yty.ctypes.data_as(ctypes.c_void_p)))
    return G, b, float(yty[0])


def dense_predict(Phi, m, C, device=None):
    """(Phi m, rowsum((Phi C) o Phi)) of an arbitrary host feature matrix on the GPU in float64 (rr_dense_predict)."""
    dev = get_device(device)
    Phi = as_float_matrix(Phi)
    N, F = Phi.shape
    m = np.ascontiguousarray(m, dtype=np.float64)
    C = np.ascontiguousarray(C, dtype=np.float64)
    if m.shape != (F,) or C.shape != (F, F):
        raise ValueError("posterior shape does not match the feature matrix")
    Ey, Vf = np.empty(N), np.empty(N)
    _check(dev.lib, dev.lib.rr_dense_predict(dev.ctx, Phi.ctypes.data_as(ctypes.c_void_p), rr_dtype(Phi.dtype), N, F,
                                             _ld(Phi), m.ctypes.data_as(ctypes.c_void_p), C.ctypes.data_as(ctypes.c_void_p),
                                             Ey.ctypes.data_as(ctypes.c_void_p), Vf.ctypes.data_as(ctypes.c_void_p)))
    return Ey, Vf


class FeatureMatrix(object):
    """Device feature matrix of a concatenated basis (rr_featmat): children put column blocks, then one Gram."""

    def __init__(self, max_rows, F, device=None):
        self.dev = get_device(device)
        self.lib = self.dev.lib
        self.F, self.max_rows = int(F), int(max_rows)
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.rr_featmat_create(self.dev.ctx, self.max_rows, self.F, ctypes.byref(h)))
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.lib.rr_featmat_destroy(h)
            except Exception:
                pass

    def begin(self, rows):
        _check(self.lib, self.lib.rr_featmat_begin(self.h, rows))

    def put_rff(self, handle, dX, lenscale, col0):
        ls, lsp, nls = _lenscale_arg(lenscale)
        _check(self.lib, self.lib.rr_featmat_put_rff(self.h, handle.h, dX.ptr, rr_dtype(dX.dtype), dX.ld, lsp, nls, col0))

    def put_linear(self, dX, onescol, col0):
        _check(self.lib, self.lib.rr_featmat_put_linear(self.h, dX.ptr, rr_dtype(dX.dtype), dX.ld, dX.shape[1],
                                                        1 if onescol else 0, col0))

    def put_fastfood(self, ff_handle, dX, lenscale, col0):
        """FastFoodRBF's Phi of the rows dX by the chain kernel (rr_fastfood16_kernel) into columns [col0, col0 + 2 d2 k)."""
        ls, lsp, nls = _lenscale_arg(lenscale)
        _check(self.lib, self.lib.rr_featmat_put_fastfood(self.h, ff_handle.h, dX.ptr, rr_dtype(dX.dtype), dX.ld, lsp, nls, col0))

    def put_fastfood_gm(self, ff_handle, dX, mean, lenscale, col0):
        """FastFoodGM's four blocks of the rows dX by the chain kernel into columns [col0, col0 + 4 d2 k)."""
        ls, lsp, nls = _lenscale_arg(lenscale)
        mu = np.ascontiguousarray(mean, dtype=np.float64)
        _check(self.lib, self.lib.rr_featmat_put_fastfood_gm(self.h, ff_handle.h, dX.ptr, rr_dtype(dX.dtype), dX.ld,
                                                             mu.ctypes.data_as(ctypes.c_void_p), lsp, nls, col0))

    def put_host(self, Phi, col0):
        Phi = as_float_matrix(Phi)
        _check(self.lib, self.lib.rr_featmat_put_host(self.h, Phi.ctypes.data_as(ctypes.c_void_p), rr_dtype(Phi.dtype),
                                                      Phi.shape[1], _ld(Phi), col0))

    def pass2_begin(self, m, C, predict=False):
        """C: host (F, F) array or a device buffer / pointer (float64, rr_posterior_dev's output).  predict: only
        predict_rows follows (C is then kept in triangular form: half the Phi C product)."""
        m = np.ascontiguousarray(m, dtype=np.float64)
        if m.shape != (self.F,):
            raise ValueError("posterior shape does not match the feature matrix")
        mp = m.ctypes.data_as(ctypes.c_void_p)
        if predict and isinstance(C, DeviceCovariance):  # variance as a sum of squares (rr_variance_factor_dev)
            if C.F != self.F:
                raise ValueError("posterior shape does not match the feature matrix")
            B, form = C.factor()
            _check(self.lib, self.lib.rr_featmat_predict_begin_b(self.h, mp, B.ptr, form))
            return
        if isinstance(C, np.ndarray):
            C = np.ascontiguousarray(C, dtype=np.float64)
            if C.shape != (self.F, self.F):
                raise ValueError("posterior shape does not match the feature matrix")
            cp = C.ctypes.data_as(ctypes.c_void_p)
            if predict:
                _check(self.lib, self.lib.rr_featmat_predict_begin(self.h, mp, cp, 0))
            else:
                _check(self.lib, self.lib.rr_featmat_pass2_begin(self.h, mp, cp))
        elif predict:
            _check(self.lib, self.lib.rr_featmat_predict_begin(self.h, mp, _ptr(C), 1))
        else:
            _check(self.lib, self.lib.rr_featmat_pass2_begin_devc(self.h, mp, _ptr(C)))

    def pass2_rows(self, dy):
        _check(self.lib, self.lib.rr_featmat_pass2_rows(self.h, _ptr(dy), rr_dtype(dy.dtype) if dy is not None else 0))

    def pass2_plan_rff(self, handle, dX, col0, dT):
        """Announce a pass2_rff call (same arguments) ahead of pass2_rows_planned."""
        _check(self.lib, self.lib.rr_featmat_pass2_plan_rff(self.h, handle.h, dX.ptr, rr_dtype(dX.dtype), dX.ld, col0, _ptr(dT)))

    def pass2_rows_planned(self, dy):
        """pass2_rows when the planned children are the only consumers of U = Phi C (which may then never be formed)."""
        _check(self.lib, self.lib.rr_featmat_pass2_rows_planned(self.h, _ptr(dy), rr_dtype(dy.dtype)))

    def pass2_rff(self, handle, dX, col0, dT):
        _check(self.lib, self.lib.rr_featmat_pass2_rff(self.h, handle.h, dX.ptr, rr_dtype(dX.dtype), dX.ld, col0, _ptr(dT)))

    def pass2_end(self):
        sq = np.zeros(1)
        _check(self.lib, self.lib.rr_featmat_pass2_end(self.h, sq.ctypes.data_as(ctypes.c_void_p)))
        return float(sq[0])

    def predict_rows(self, rows):
        Ey, Vf = np.empty(rows), np.empty(rows)
        _check(self.lib, self.lib.rr_featmat_predict_rows(self.h, Ey.ctypes.data_as(ctypes.c_void_p),
                                                          Vf.ctypes.data_as(ctypes.c_void_p)))
        return Ey, Vf

    def glm_step(self, dy, drowarg, lik, lik_param, WS, K, L):
        """(Edws (K*L, F), llsum (K,), aux (K,)) of one SVI minibatch step (rr_featmat_glm_step)."""
        WS = np.ascontiguousarray(WS, dtype=np.float64)
        if WS.shape != (K * L, self.F):
            raise ValueError("weight samples must have shape (K*L, F)")
        Edws, ll, aux = np.empty((K * L, self.F)), np.empty(K), np.empty(K)
        _check(self.lib, self.lib.rr_featmat_glm_step(self.h, _ptr(dy), _ptr(drowarg), rr_dtype(dy.dtype), int(lik),
                                                      float(lik_param), WS.ctypes.data_as(ctypes.c_void_p), K, L,
                                                      Edws.ctypes.data_as(ctypes.c_void_p),
                                                      ll.ctypes.data_as(ctypes.c_void_p),
                                                      aux.ctypes.data_as(ctypes.c_void_p)))
        return Edws, ll, aux

    def glm_step_sampled(self, dy, drowarg, lik, lik_param, m, C, K, L, seed, step, objective_only=False):
        """(Edm (F, K), EdC (F, K), llsum (K,), aux (K,)) with the reparameterisation draws made on the device;
        objective_only: (None, None, llsum, aux) without the gradient GEMMs."""
        m = np.ascontiguousarray(m, dtype=np.float64)
        C = np.ascontiguousarray(C, dtype=np.float64)
        if m.shape != (self.F, K) or C.shape != (self.F, K):
            raise ValueError("m and C must have shape (F, K)")
        Edm, EdC, ll, aux = np.empty((K, self.F)), np.empty((K, self.F)), np.empty(K), np.empty(K)
        none = ctypes.c_void_p(None)
        _check(self.lib, self.lib.rr_featmat_glm_step_sampled(
            self.h, _ptr(dy), _ptr(drowarg), rr_dtype(dy.dtype), int(lik), float(lik_param),
            m.ctypes.data_as(ctypes.c_void_p), C.ctypes.data_as(ctypes.c_void_p), K, L, int(seed), int(step),
            none if objective_only else Edm.ctypes.data_as(ctypes.c_void_p),
            none if objective_only else EdC.ctypes.data_as(ctypes.c_void_p), ll.ctypes.data_as(ctypes.c_void_p),
            aux.ctypes.data_as(ctypes.c_void_p)))
        return (None, None, ll, aux) if objective_only else (Edm.T, EdC.T, ll, aux)

    def glm_step_draws(self, dy, drowarg, lik, lik_param, m, C, K, L, E, objective_only=False):
        """(Edm (F, K), EdC (F, K), llsum, aux) for the caller's standard-normal draws E (K*L, F); objective_only as in
        glm_step_sampled."""
        m = np.ascontiguousarray(m, dtype=np.float64)
        C = np.ascontiguousarray(C, dtype=np.float64)
        on_device = isinstance(E, DeviceBuffer)  # float32 (K*L, F) uploaded ahead of the step (rr_featmat_glm_step_draws_dev)
        if not on_device:
            E = np.ascontiguousarray(E, dtype=np.float32)
        if m.shape != (self.F, K) or C.shape != (self.F, K) or tuple(E.shape) != (K * L, self.F) or E.dtype != np.float32:
            raise ValueError("m, C must have shape (F, K) and E (K*L, F)")
        Edm, EdC, ll, aux = np.empty((K, self.F)), np.empty((K, self.F)), np.empty(K), np.empty(K)
        none = ctypes.c_void_p(None)
        step = self.lib.rr_featmat_glm_step_draws_dev if on_device else self.lib.rr_featmat_glm_step_draws
        _check(self.lib, step(
            self.h, _ptr(dy), _ptr(drowarg), rr_dtype(dy.dtype), int(lik), float(lik_param),
            m.ctypes.data_as(ctypes.c_void_p), C.ctypes.data_as(ctypes.c_void_p), K, L,
            E.ptr if on_device else E.ctypes.data_as(ctypes.c_void_p),
            none if objective_only else Edm.ctypes.data_as(ctypes.c_void_p),
            none if objective_only else EdC.ctypes.data_as(ctypes.c_void_p), ll.ctypes.data_as(ctypes.c_void_p),
            aux.ctypes.data_as(ctypes.c_void_p)))
        return (None, None, ll, aux) if objective_only else (Edm.T, EdC.T, ll, aux)

    def glm_plan_rff(self, handle, dX, col0, dT):
        """Announce the glm_rff call that follows the next step (the step may then contract EdPhi itself)."""
        _check(self.lib, self.lib.rr_featmat_glm_plan_rff(self.h, handle.h, dX.ptr, rr_dtype(dX.dtype), dX.ld, col0, _ptr(dT)))

    def glm_rff(self, handle, dX, col0, dT):
        _check(self.lib, self.lib.rr_featmat_glm_rff(self.h, handle.h, dX.ptr, rr_dtype(dX.dtype), dX.ld, col0, _ptr(dT)))

    def glm_edphi(self, rows, col0, ncols):
        E = np.empty((rows, ncols))
        _check(self.lib, self.lib.rr_featmat_glm_edphi(self.h, col0, ncols, E.ctypes.data_as(ctypes.c_void_p)))
        return E

    def project(self, rows, W):
        """(rows, S) = P W for a host (F, S) matrix (rr_featmat_project)."""
        W = np.ascontiguousarray(W, dtype=np.float64)
        if W.ndim != 2 or W.shape[0] != self.F:
            raise ValueError("W must have shape (F, S)")
        out = np.empty((rows, W.shape[1]))
        _check(self.lib, self.lib.rr_featmat_project(self.h, W.ctypes.data_as(ctypes.c_void_p), W.shape[1],
                                                     out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def gram_into(self, dy, dG, db=None, dyty=None):
        _check(self.lib, self.lib.rr_featmat_gram(self.h, _ptr(dy), rr_dtype(dy.dtype) if dy is not None else 0,
                                                  _ptr(dG), _ptr(db), _ptr(dyty)))


UPDATER_IDS = {"SGDUpdater": 0, "AdaDelta": 1, "AdaGrad": 2, "Momentum": 3, "Adam": 4}   # RR_UPD_*


class SgdChild(ctypes.Structure):
    """rr_glm_sgd_child (include/revrand_hip.h)"""
    _fields_ = [("kind", ctypes.c_int), ("basis", ctypes.c_void_p), ("d", ctypes.c_int), ("onescol", ctypes.c_int),
                ("n_ls", ctypes.c_int)]


class ResidentSgd(object):
    """The SVI loop with its parameters in HBM (rr_glm_sgd): `step` queues one whole SGD step and returns at once.
    children: ("rff", RffHandle, n_ls) | ("gm", RffHandle of the chain's dense equivalent, 2 Xdim) | ("linear", d, onescol) in
    concatenation order."""

    def __init__(self, fm, children, K, n_lik, z0, lower, upper, is_log, updater_id, updater_par, maxiter):
        self.fm, self.children, self.lib = fm, list(children), fm.lib      # (all kept alive for as long as the loop)
        z0 = np.ascontiguousarray(z0, dtype=np.float64)
        lower = np.ascontiguousarray(lower, dtype=np.float64)
        upper = np.ascontiguousarray(upper, dtype=np.float64)
        is_log = np.ascontiguousarray(is_log, dtype=np.uint8)
        kids = (SgdChild * len(self.children))()
        n_ls = 0
        for k, ch in zip(kids, self.children):
            if ch[0] in ("rff", "gm"):   # gm: a spectral-mixture component on its dense handle, [mean | length scales]
                k.kind, k.basis, k.d, k.onescol, k.n_ls = (0 if ch[0] == "rff" else 2), ch[1].h, 0, 0, int(ch[2])
                n_ls += int(ch[2])
            else:
                k.kind, k.basis, k.d, k.onescol, k.n_ls = 1, None, int(ch[1]), 1 if ch[2] else 0, 0
        self.nk = len(self.children)
        self.np_ = 2 * fm.F * K + self.nk + n_lik + n_ls
        if not (z0.shape == lower.shape == upper.shape == is_log.shape == (self.np_,)):
            raise ValueError("z0, lower, upper, is_log must have 2 F K + children + n_lik + length scales = %d entries" % self.np_)
        par = np.zeros(4)
        par[:len(updater_par)] = updater_par
        self.maxiter = int(maxiter)
        self._ptrs, self._dts, self._lds = (ctypes.c_void_p * self.nk)(), (ctypes.c_int * self.nk)(), (ctypes.c_int64 * self.nk)()
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.rr_glm_sgd_create(fm.h, self.nk, ctypes.cast(kids, ctypes.c_void_p), K, n_lik,
                                                    z0.ctypes.data_as(ctypes.c_void_p), lower.ctypes.data_as(ctypes.c_void_p),
                                                    upper.ctypes.data_as(ctypes.c_void_p), is_log.ctypes.data_as(ctypes.c_void_p),
                                                    int(updater_id), par.ctypes.data_as(ctypes.c_void_p), self.maxiter,
                                                    ctypes.byref(h)))
        self.h = h

    def step(self, dXs, rows, dy, drowarg, lik, llconst, bmag, L, dE=None, seed=0, key=0, comm=None):
        """dXs: every child's minibatch rows (DeviceView), in concatenation order.  comm (an rr_comm handle of this rank, one
        process per GPU): rr_glm_sgd_dist_step -- the step's row sums all-reduced over the ranks in HBM."""
        for i, dX in enumerate(dXs):
            p = dX.ptr
            self._ptrs[i] = p if isinstance(p, int) else p.value
            self._dts[i], self._lds[i] = rr_dtype(dX.dtype), dX.ld
        args = (ctypes.cast(self._ptrs, ctypes.c_void_p), ctypes.cast(self._dts, ctypes.c_void_p),
                ctypes.cast(self._lds, ctypes.c_void_p), int(rows), _ptr(dy), _ptr(drowarg),
                rr_dtype(dy.dtype), int(lik), float(llconst), float(bmag), int(L),
                None if dE is None else dE.ptr, int(seed), int(key))
        if comm is None:
            _check(self.lib, self.lib.rr_glm_sgd_step(self.h, *args))
        else:
            _check(self.lib, self.lib.rr_glm_sgd_dist_step(self.h, comm, *args))

    def objective(self, step):
        v = ctypes.c_double()
        _check(self.lib, self.lib.rr_glm_sgd_objective(self.h, int(step), ctypes.byref(v)))
        return v.value

    def read(self):
        """(z, objs, norms) after waiting for every queued step."""
        z, objs, norms = np.empty(self.np_), np.empty(self.maxiter), np.empty(self.maxiter)
        n = ctypes.c_int64()
        _check(self.lib, self.lib.rr_glm_sgd_read(self.h, z.ctypes.data_as(ctypes.c_void_p), objs.ctypes.data_as(ctypes.c_void_p),
                                                  norms.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)))
        return z, objs[:n.value], norms[:n.value]

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.lib.rr_glm_sgd_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SgdBatch(ctypes.Structure):
    """rr_glm_sgd_batch (include/revrand_hip.h)"""
    _fields_ = [("dX", ctypes.c_void_p), ("x_dtype", ctypes.c_void_p), ("ldx", ctypes.c_void_p), ("rows", ctypes.c_int64),
                ("dy", ctypes.c_void_p), ("drowarg", ctypes.c_void_p), ("dE", ctypes.c_void_p)]


class ResidentSgdGroup(object):
    """One ResidentSgd per member of a device group, stepped together (rr_glm_sgd_group_step): every member works on ITS rows of
    the minibatch, the row sums are all-reduced in HBM, the members' copies of the parameters stay bit-identical."""

    def __init__(self, comms, sgds):
        self.comms, self.sgds, self.n, self.lib = comms, list(sgds), len(sgds), sgds[0].lib
        self._loops = (ctypes.c_void_p * self.n)(*[s.h for s in self.sgds])
        self._batches = (SgdBatch * self.n)()

    def step(self, parts, lik, llconst, bmag, L, seed=0, key=0):
        """parts[i] = (dXs, rows, dy, drowarg, dE) of member i (rows == 0: the rest may be None)."""
        dtype = None
        for b, sgd, (dXs, rows, dy, drowarg, dE) in zip(self._batches, self.sgds, parts):
            b.rows = int(rows)
            if not rows:
                b.dX = b.x_dtype = b.ldx = b.dy = b.drowarg = b.dE = None
                continue
            for i, dX in enumerate(dXs):
                p = dX.ptr
                sgd._ptrs[i] = p if isinstance(p, int) else p.value
                sgd._dts[i], sgd._lds[i] = rr_dtype(dX.dtype), dX.ld
            b.dX, b.x_dtype, b.ldx = (ctypes.cast(a, ctypes.c_void_p) for a in (sgd._ptrs, sgd._dts, sgd._lds))
            b.dy, b.drowarg, b.dE = _ptr(dy), _ptr(drowarg), None if dE is None else dE.ptr
            dtype = rr_dtype(dy.dtype)
        _check(self.lib, self.lib.rr_glm_sgd_group_step(self.n, ctypes.cast(self._loops, ctypes.c_void_p), self.comms,
                                                        ctypes.cast(self._batches, ctypes.c_void_p), dtype, int(lik), float(llconst),
                                                        float(bmag), int(L), int(seed), int(key)))

    def objective(self, step):
        return self.sgds[0].objective(step)

    def read(self, member=0):
        return self.sgds[member].read()

    def close(self):
        for s in self.sgds:
            s.close()


class FusedSvi(object):
    """The SVI loop for small minibatches, many steps per launch (rr_glm_svi, rr_svi.hip).
    children: ("rff", RffHandle, n_ls, dX) | ("linear", d, onescol, dX) in concatenation order, dX the child's RESIDENT
    rows (DeviceMatrix, all N rows); dy / drowarg: DeviceBuffers of all N targets / per-row arguments."""

    def __init__(self, dev, children, N, dy, drowarg, dlconst, K, L, M, lik, n_lik, z0, lower, upper, is_log, updater_id, updater_par,
                 maxiter, bmag):
        self.dev, self.lib, self.children = dev, dev.lib, list(children)
        self._keep = (dy, drowarg, dlconst)
        z0 = np.ascontiguousarray(z0, dtype=np.float64)
        lower = np.ascontiguousarray(lower, dtype=np.float64)
        upper = np.ascontiguousarray(upper, dtype=np.float64)
        is_log = np.ascontiguousarray(is_log, dtype=np.uint8)
        nk = len(self.children)
        kids = (SgdChild * nk)()
        ptrs, dts, lds = (ctypes.c_void_p * nk)(), (ctypes.c_int * nk)(), (ctypes.c_int64 * nk)()
        F = n_ls = 0
        for i, (k, ch) in enumerate(zip(kids, self.children)):
            dX = ch[3]
            if ch[0] == "rff":
                k.kind, k.basis, k.d, k.onescol, k.n_ls = 0, ch[1].h, 0, 0, int(ch[2])
                n_ls += int(ch[2])
                F += 2 * ch[1].n
            else:
                k.kind, k.basis, k.d, k.onescol, k.n_ls = 1, None, int(ch[1]), 1 if ch[2] else 0, 0
                F += int(ch[1]) + (1 if ch[2] else 0)
            p = dX.ptr
            ptrs[i] = p if isinstance(p, int) else p.value
            dts[i], lds[i] = rr_dtype(dX.dtype), dX.ld
        self.F, self.K, self.L, self.M = F, int(K), int(L), int(M)
        self.np_ = 2 * F * self.K + nk + n_lik + n_ls
        if not (z0.shape == lower.shape == upper.shape == is_log.shape == (self.np_,)):
            raise ValueError("z0, lower, upper, is_log must have 2 F K + children + n_lik + length scales = %d entries" % self.np_)
        par = np.zeros(4)
        par[:len(updater_par)] = updater_par
        self.maxiter = int(maxiter)
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.rr_glm_svi_create(
            dev.ctx, nk, ctypes.cast(kids, ctypes.c_void_p), ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(dts, ctypes.c_void_p),
            ctypes.cast(lds, ctypes.c_void_p), int(N), _ptr(dy), _ptr(drowarg), _ptr(dlconst), rr_dtype(dy.dtype), self.K, self.L, self.M,
            int(lik), int(n_lik), z0.ctypes.data_as(ctypes.c_void_p), lower.ctypes.data_as(ctypes.c_void_p),
            upper.ctypes.data_as(ctypes.c_void_p), is_log.ctypes.data_as(ctypes.c_void_p), int(updater_id),
            par.ctypes.data_as(ctypes.c_void_p), self.maxiter, float(bmag), ctypes.byref(h)))
        self.h = h

    def set_start(self, z0, lower, upper, is_log):
        arrs = [np.ascontiguousarray(z0, dtype=np.float64), np.ascontiguousarray(lower, dtype=np.float64),
                np.ascontiguousarray(upper, dtype=np.float64), np.ascontiguousarray(is_log, dtype=np.uint8)]
        if any(a.shape != (self.np_,) for a in arrs):
            raise ValueError("set_start: %d coordinates expected" % self.np_)
        _check(self.lib, self.lib.rr_glm_svi_set_start(self.h, *[a.ctypes.data_as(ctypes.c_void_p) for a in arrs]))

    def run(self, steps, didx, dE=None, seed=0, key0=0):
        """`steps` SGD steps in one launch (asynchronous): didx a device int32 (steps, M) buffer, dE device float32 draws or None."""
        _check(self.lib, self.lib.rr_glm_svi_run(self.h, int(steps), _ptr(didx), None if dE is None else _ptr(dE), int(seed), int(key0)))

    def starts(self, didx, cand, dE=None, seed=0, key0=0):
        """-ELBO of every candidate row of `cand` (ncand, np; x space) on its own minibatch / draws: one launch."""
        cand = np.ascontiguousarray(cand, dtype=np.float64)
        out = np.empty(cand.shape[0])
        _check(self.lib, self.lib.rr_glm_svi_starts(self.h, cand.shape[0], _ptr(didx), cand.ctypes.data_as(ctypes.c_void_p),
                                                    None if dE is None else _ptr(dE), int(seed), int(key0),
                                                    out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def read(self):
        z, objs, norms = np.empty(self.np_), np.empty(self.maxiter), np.empty(self.maxiter)
        n = ctypes.c_int64()
        _check(self.lib, self.lib.rr_glm_svi_read(self.h, z.ctypes.data_as(ctypes.c_void_p), objs.ctypes.data_as(ctypes.c_void_p),
                                                  norms.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)))
        return z, objs[:n.value], norms[:n.value]

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.lib.rr_glm_svi_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def svi_supported(F, K, L, M, n_children, dsum, n_ls):
    return bool(load_library().rr_glm_svi_supported(int(F), int(K), int(L), int(M), int(n_children), int(dsum), int(n_ls)))


class FeatureMatrix64(object):
    """The float64 device feature matrix of a concatenated basis (rr_featmat64): the subset of FeatureMatrix a resident fit
    needs -- children put column blocks, Gram, second pass, prediction -- in the reference's arithmetic."""

    dtype = np.dtype(np.float64)

    def __init__(self, max_rows, F, device=None):
        self.dev = get_device(device)
        self.lib = self.dev.lib
        self.F, self.max_rows = int(F), int(max_rows)
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.rr_featmat64_create(self.dev.ctx, self.max_rows, self.F, ctypes.byref(h)))
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.lib.rr_featmat64_destroy(h)
            except Exception:
                pass

    def begin(self, rows):
        _check(self.lib, self.lib.rr_featmat64_begin(self.h, rows))

    def put_rff(self, handle, dX, lenscale, col0):
        ls, lsp, nls = _lenscale_arg(lenscale)
        _check(self.lib, self.lib.rr_featmat64_put_rff(self.h, handle.h, dX.ptr, rr_dtype(dX.dtype), dX.ld, lsp, nls, col0))

    def put_linear(self, dX, onescol, col0):
        _check(self.lib, self.lib.rr_featmat64_put_linear(self.h, dX.ptr, rr_dtype(dX.dtype), dX.ld, dX.shape[1],
                                                          1 if onescol else 0, col0))

    def put_host(self, Phi, col0):
        Phi = as_float_matrix(Phi)
        _check(self.lib, self.lib.rr_featmat64_put_host(self.h, Phi.ctypes.data_as(ctypes.c_void_p), rr_dtype(Phi.dtype),
                                                        Phi.shape[1], _ld(Phi), col0))

    def gram_into(self, dy, dG, db=None, dyty=None):
        _check(self.lib, self.lib.rr_featmat64_gram(self.h, _ptr(dy), rr_dtype(dy.dtype) if dy is not None else 0,
                                                    _ptr(dG), _ptr(db), _ptr(dyty)))

    def pass2_begin(self, m, C, predict=False):
        """C: host (F, F) array or a device buffer / pointer (float64).  (`predict` is accepted for symmetry: the float64
        product is always the full one.)"""
        m = np.ascontiguousarray(m, dtype=np.float64)
        if m.shape != (self.F,):
            raise ValueError("posterior shape does not match the feature matrix")
        mp = m.ctypes.data_as(ctypes.c_void_p)
        if isinstance(C, np.ndarray):
            C = np.ascontiguousarray(C, dtype=np.float64)
            if C.shape != (self.F, self.F):
                raise ValueError("posterior shape does not match the feature matrix")
            _check(self.lib, self.lib.rr_featmat64_pass2_begin(self.h, mp, C.ctypes.data_as(ctypes.c_void_p), 0))
        else:
            _check(self.lib, self.lib.rr_featmat64_pass2_begin(self.h, mp, _ptr(C), 1))

    def pass2_rows(self, dy):
        _check(self.lib, self.lib.rr_featmat64_pass2_rows(self.h, _ptr(dy), rr_dtype(dy.dtype)))

    def pass2_rff(self, handle, dX, col0, dT):
        _check(self.lib, self.lib.rr_featmat64_pass2_rff(self.h, handle.h, dX.ptr, rr_dtype(dX.dtype), dX.ld, col0, _ptr(dT)))

    def pass2_end(self):
        sq = np.zeros(1)
        _check(self.lib, self.lib.rr_featmat64_pass2_end(self.h, sq.ctypes.data_as(ctypes.c_void_p)))
        return float(sq[0])

    def predict_rows(self, rows):
        Ey, Vf = np.empty(rows), np.empty(rows)
        _check(self.lib, self.lib.rr_featmat64_predict_rows(self.h, Ey.ctypes.data_as(ctypes.c_void_p),
                                                            Vf.ctypes.data_as(ctypes.c_void_p)))
        return Ey, Vf


def hadamard(Y, ordering=True, device=None):
    """Row-wise Walsh-Hadamard transform on the GPU (rr_hadamard)."""
    dev = get_device(device)
    Y = np.ascontiguousarray(Y)
    if Y.dtype not in (np.float32, np.float64):
        Y = Y.astype(np.float64)
    if Y.ndim != 2:
        raise ValueError("expected a 2-d array")
    out = np.empty_like(Y)
    _check(dev.lib, dev.lib.rr_hadamard(dev.ctx, Y.ctypes.data_as(ctypes.c_void_p), rr_dtype(Y.dtype), Y.shape[0],
                                        Y.shape[1], 1 if ordering else 0, out.ctypes.data_as(ctypes.c_void_p)))
    return out


class FastFoodHandle(object):
    """Device-resident FastFood block matrices (rr_basis of kind FASTFOOD)."""

    def __init__(self, d, d2, k, B, G, PI, S, compute="f32", device=None):
        self.dev = get_device(device)
        self.lib = self.dev.lib
        self.d, self.d2, self.k, self.n = d, d2, k, d2 * k
        B = np.ascontiguousarray(B, dtype=np.int64)
        PI = np.ascontiguousarray(PI, dtype=np.int64)
        G = np.ascontiguousarray(G, dtype=np.float64)
        S = np.ascontiguousarray(S, dtype=np.float64)
        if not (B.shape == G.shape == PI.shape == S.shape == (k, d2)):
            raise ValueError("FastFood matrices must all have shape (k, d2)")
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.rr_fastfood_create(self.dev.ctx, {"f32": RR_F32, "f64": RR_F64}[compute], d, d2, k,
                                                     B.ctypes.data_as(ctypes.c_void_p),
                                                     G.ctypes.data_as(ctypes.c_void_p),
                                                     PI.ctypes.data_as(ctypes.c_void_p),
                                                     S.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)))
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.lib.rr_basis_destroy(h)
            except Exception:
                pass

    def _call(self, fn, X, lenscale, width, out_dtype):
        X = as_float_matrix(X)
        N = X.shape[0]
        out = np.empty((N, width), dtype=out_dtype)
        ls, lsp, nls = _lenscale_arg(lenscale)
        _check(self.lib, fn(self.h, X.ctypes.data_as(ctypes.c_void_p), rr_dtype(X.dtype), N, _ld(X), lsp, nls,
                            out.ctypes.data_as(ctypes.c_void_p), rr_dtype(out.dtype), width))
        return out

    def transform(self, X, lenscale, out_dtype=np.float64):
        return self._call(self.lib.rr_fastfood_transform, X, lenscale, 2 * self.n, out_dtype)

    def transform_dev(self, dX, lenscale, dOut, out_dtype=np.float32, ldphi=None):
        """Device-resident X (DeviceMatrix) -> Phi in the device buffer dOut (asynchronous)."""
        ls, lsp, nls = _lenscale_arg(lenscale)
        _check(self.lib, self.lib.rr_fastfood_transform_dev(self.h, dX.ptr, rr_dtype(dX.dtype), dX.shape[0], dX.ld, lsp, nls,
                                                            _ptr(dOut), rr_dtype(np.dtype(out_dtype)),
                                                            2 * self.n if ldphi is None else ldphi))

    def vx(self, X, lenscale=1.0, out_dtype=np.float64):
        return self._call(self.lib.rr_fastfood_vx, X, lenscale, self.n, out_dtype)

    def gm_transform_dev(self, dX, mean, lenscale, dOut, out_dtype=np.float32, ldphi=None):
        """Device-resident X (DeviceMatrix) -> the mixture component's (rows, 4n) features in dOut (asynchronous)."""
        ls, lsp, nls = _lenscale_arg(lenscale)
        mu = np.ascontiguousarray(mean, dtype=np.float64)
        _check(self.lib, self.lib.rr_fastfood_gm_transform_dev(self.h, dX.ptr, rr_dtype(dX.dtype), dX.shape[0], dX.ld,
                                                               mu.ctypes.data_as(ctypes.c_void_p), lsp, nls, _ptr(dOut),
                                                               rr_dtype(np.dtype(out_dtype)), 4 * self.n if ldphi is None else ldphi))

    @property
    def gm_chain_ok(self):
        """The chain kernels' mixture-component mode serves this block size (16 <= d2 <= 256)."""
        return 16 <= self.d2 <= 256

    def gm_transform(self, X, mean, lenscale, out_dtype=np.float64):
        """FastFoodGM.transform by the chain kernel: (N, 4n) = [cos | sin](VX + mX), [cos | sin](VX - mX), / sqrt(2n)."""
        X = as_float_matrix(X)
        N = X.shape[0]
        out = np.empty((N, 4 * self.n), dtype=out_dtype)
        ls, lsp, nls = _lenscale_arg(lenscale)
        mu = np.ascontiguousarray(mean, dtype=np.float64)
        _check(self.lib, self.lib.rr_fastfood_gm_transform(self.h, X.ctypes.data_as(ctypes.c_void_p), rr_dtype(X.dtype), N, _ld(X),
                                                           mu.ctypes.data_as(ctypes.c_void_p), lsp, nls,
                                                           out.ctypes.data_as(ctypes.c_void_p), rr_dtype(out.dtype), 4 * self.n))
        return out


class RffHandle(object):
    """Device-resident random Fourier basis (rr_basis): W lives on the GPU."""

    def __init__(self, W, compute="f32", device=None):
        self.dev = get_device(device)
        self.lib = self.dev.lib
        W = np.ascontiguousarray(W, dtype=np.float64)
        self.d, self.n = W.shape
        # "f32p64" (RR_F32P64): the f32 pipeline with the phases accumulated and reduced in float64 -- heavy-tailed W
        code = {"f32": RR_F32, "f64": RR_F64, "f32p64": RR_F32P64}[compute]
        self.phase64 = code == RR_F32P64
        self.compute = RR_F32 if self.phase64 else code  # arithmetic of every product behind the feature kernel
        # what a resident X (and the y next to it) is kept in: float64 unless the whole pipeline is float32
        self.x_dtype = np.dtype(np.float32 if code == RR_F32 else np.float64)
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.rr_rff_create(self.dev.ctx, code, self.d, self.n,
                                                W.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)))
        self.h = h
        self.padded_dim = self.lib.rr_rff_padded_dim(h)

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.lib.rr_basis_destroy(h)
            except Exception:
                pass

    def transform(self, X, lenscale, out_dtype=np.float64):
        X = as_float_matrix(X)
        N = X.shape[0]
        out = np.empty((N, 2 * self.n), dtype=out_dtype)
        ls, lsp, nls = _lenscale_arg(lenscale)
        _check(self.lib, self.lib.rr_rff_transform(self.h, X.ctypes.data_as(ctypes.c_void_p), rr_dtype(X.dtype), N,
                                                   _ld(X), lsp, nls, out.ctypes.data_as(ctypes.c_void_p),
                                                   rr_dtype(out.dtype), 2 * self.n))
        return out

    def grad(self, X, lenscale, out_dtype=np.float64):
        X = as_float_matrix(X)
        N = X.shape[0]
        ls, lsp, nls = _lenscale_arg(lenscale)
        shape = (N, 2 * self.n) if nls == 1 else (N, 2 * self.n, self.d)
        out = np.empty(shape, dtype=out_dtype)
        _check(self.lib, self.lib.rr_rff_grad(self.h, X.ctypes.data_as(ctypes.c_void_p), rr_dtype(X.dtype), N,
                                              _ld(X), lsp, nls, out.ctypes.data_as(ctypes.c_void_p),
                                              rr_dtype(out.dtype)))
        return out

    def gram(self, X, y, lenscale):
        """(G (F,F) full symmetric, b (F,), yty) from host X (N,d), y (N,) or None."""
        X = as_float_matrix(X)
        N = X.shape[0]
        F = 2 * self.n
        G = np.empty((F, F))
        ls, lsp, nls = _lenscale_arg(lenscale)
        if y is None:
            _check(self.lib, self.lib.rr_rff_gram(self.h, X.ctypes.data_as(ctypes.c_void_p), None, rr_dtype(X.dtype),
                                                  N, _ld(X), lsp, nls, G.ctypes.data_as(ctypes.c_void_p), None, None))
            return G, None, None
        y = np.ascontiguousarray(y, dtype=X.dtype).ravel()
        if y.shape[0] != N:
            raise ValueError("X and y have inconsistent numbers of rows")
        b = np.empty(F)
        yty = np.empty(1)
        _check(self.lib, self.lib.rr_rff_gram(self.h, X.ctypes.data_as(ctypes.c_void_p),
                                              y.ctypes.data_as(ctypes.c_void_p), rr_dtype(X.dtype), N, _ld(X), lsp,
                                              nls, G.ctypes.data_as(ctypes.c_void_p),
                                              b.ctypes.data_as(ctypes.c_void_p), yty.ctypes.data_as(ctypes.c_void_p)))
        return G, b, float(yty[0])

    def grad_contract(self, X, E, lenscale):
        """T (d, n) with  sum(E o dPhi_i) = -(1/l_i^2) sum_f W[i,f] T[i,f]  (rr_rff_grad_contract)."""
        X = as_float_matrix(X)
        E = as_float_matrix(E)
        if E.shape != (X.shape[0], 2 * self.n):
            raise ValueError("E must have shape (N, 2*nbases)")
        ls, lsp, nls = _lenscale_arg(lenscale)
        T = np.zeros((self.d, self.n))
        _check(self.lib, self.lib.rr_rff_grad_contract(self.h, X.ctypes.data_as(ctypes.c_void_p), rr_dtype(X.dtype),
                                                       X.shape[0], _ld(X), lsp, nls, E.ctypes.data_as(ctypes.c_void_p),
                                                       rr_dtype(E.dtype), _ld(E), T.ctypes.data_as(ctypes.c_void_p)))
        return T

    def gm_transform(self, X, mean, lenscale, out_dtype=np.float64):
        """Spectral-mixture features (N, 4n) on this handle's W (rr_gm_transform)."""
        X = as_float_matrix(X)
        N = X.shape[0]
        out = np.empty((N, 4 * self.n), dtype=out_dtype)
        ls, lsp, nls = _lenscale_arg(lenscale)
        mu = np.ascontiguousarray(mean, dtype=np.float64)
        _check(self.lib, self.lib.rr_gm_transform(self.h, X.ctypes.data_as(ctypes.c_void_p), rr_dtype(X.dtype), N, _ld(X),
                                                  mu.ctypes.data_as(ctypes.c_void_p), lsp, nls,
                                                  out.ctypes.data_as(ctypes.c_void_p), rr_dtype(out.dtype), 4 * self.n))
        return out

    def gm_grad(self, X, mean, lenscale, out_dtype=np.float64):
        X = as_float_matrix(X)
        N = X.shape[0]
        ls, lsp, nls = _lenscale_arg(lenscale)
        mu = np.ascontiguousarray(mean, dtype=np.float64)
        shape = (N, 4 * self.n) if self.d == 1 else (N, 4 * self.n, self.d)
        dm, dl = np.empty(shape, dtype=out_dtype), np.empty(shape, dtype=out_dtype)
        _check(self.lib, self.lib.rr_gm_grad(self.h, X.ctypes.data_as(ctypes.c_void_p), rr_dtype(X.dtype), N, _ld(X),
                                             mu.ctypes.data_as(ctypes.c_void_p), lsp, nls,
                                             dm.ctypes.data_as(ctypes.c_void_p), dl.ctypes.data_as(ctypes.c_void_p),
                                             rr_dtype(dm.dtype)))
        return dm, dl

    # -- device-resident API (fit loops, bench, shards) -------------------------
    def upload(self, X):
        """Upload X once in the padded layout the kernels read (rr_rff_padded_dim)."""
        return self.dev.upload_matrix(X, ld_dev=self.padded_dim)

    def gram_dev(self, dX, dy, lenscale, dG, db=None, dyty=None):
        """Accumulate into device buffers (async).  dX: DeviceMatrix; dy: DeviceBuffer or None."""
        ls, lsp, nls = _lenscale_arg(lenscale)
        _check(self.lib, self.lib.rr_rff_gram_dev(self.h, dX.ptr, _ptr(dy), rr_dtype(dX.dtype), dX.shape[0],
                                                  dX.ld, lsp, nls, _ptr(dG), _ptr(db), _ptr(dyty)))

    def gram_host(self, dX, dy, lenscale):
        """(G, b, yty) as host arrays from device-resident X (DeviceMatrix) and y (DeviceBuffer)."""
        F = 2 * self.n
        acc = self.dev.zeros((F * F + F + 1) * 8)
        base = acc.ptr.value
        self.gram_dev(dX, dy, lenscale, ctypes.c_void_p(base), ctypes.c_void_p(base + F * F * 8),
                      ctypes.c_void_p(base + (F * F + F) * 8))
        self.symmetrize_dev(ctypes.c_void_p(base))
        out = self.dev.download(acc, (F * F + F + 1,), np.float64)
        acc.free()
        return out[:F * F].reshape(F, F), out[F * F:F * F + F].copy(), float(out[-1])

    def elbo_pass2(self, dX, dy, lenscale, m, C):
        """(sqErr, T (d, n)) for the posterior (m, C): the second data pass of `_elbo`.  C: host (F, F) array, or a
        DeviceBuffer / pointer holding it in float64 on the device (rr_posterior_dev's output)."""
        ls, lsp, nls = _lenscale_arg(lenscale)
        m = np.ascontiguousarray(m, dtype=np.float64)
        sq = np.zeros(1)
        T = np.zeros((self.d, self.n))
        if isinstance(C, np.ndarray):
            C = np.ascontiguousarray(C, dtype=np.float64)
            fn, cp = self.lib.rr_rff_elbo_pass2_dev, C.ctypes.data_as(ctypes.c_void_p)
        else:
            fn, cp = self.lib.rr_rff_elbo_pass2_devc, _ptr(C)
        _check(self.lib, fn(self.h, dX.ptr, dy.ptr, rr_dtype(dX.dtype), dX.shape[0], dX.ld, lsp, nls,
                            m.ctypes.data_as(ctypes.c_void_p), cp, sq.ctypes.data_as(ctypes.c_void_p),
                            T.ctypes.data_as(ctypes.c_void_p)))
        return float(sq[0]), T

    PREDICT_CONCURRENT_CHECK_ROWS = 32768   # from this many query rows on, `predict` validates them on a second host thread

    def predict(self, X, lenscale, m, C, check_rows=None):
        """(Ey, Vf) = (Phi m, rowsum((Phi C) o Phi)) for host query rows X.  C: host (F, F) array, or a DeviceBuffer
        holding it in float64 on the device (uploaded once by the estimator).  C = None: the mean alone, (Ey, None), from
        the feature kernel without the N x F x F product (`mean_only_ok` bases).

        `check_rows`: the caller's validation of the rows (sklearn's `check_array`: finiteness), run by this method instead
        of before it: for a large query on a host thread WHILE the rows are uploaded and the GPU works on them (3-4 ms for
        300 000 x 32 values, as much as their upload); its exception is raised before anything is returned.  (Also tried in
        round 4 and removed: the query in four row chunks, chunk k + 1 validated and uploaded under chunk k's product --
        53.7 ms against 47.9 ms per 300 000-row call: the per-call host work of four product calls and their shorter
        launches cost more than the hidden copy saves.)"""
        ls, lsp, nls = _lenscale_arg(lenscale)
        m = np.ascontiguousarray(m, dtype=np.float64)
        N = X.shape[0]
        f32_factor = isinstance(C, DeviceCovariance) and self.compute == RR_F32
        checker = None
        if check_rows is not None and N >= self.PREDICT_CONCURRENT_CHECK_ROWS:
            failed = []

            def run_check():
                try:
                    check_rows(X)
                except BaseException as e:
                    failed.append(e)
            checker = threading.Thread(target=run_check, name="rr-predict-check", daemon=True)
            checker.start()
        elif check_rows is not None:
            check_rows(X)
        try:
            return self._predict(X, N, lsp, nls, m, C, f32_factor)
        finally:
            if checker is not None:
                checker.join()
                if failed:
                    raise failed[0]

    @property
    def mean_only_ok(self):
        """rr_rff_predict_mean_dev serves this basis (float32 arithmetic and phases, Xdim <= 128)."""
        return self.compute == RR_F32 and not self.phase64 and self.d <= 128

    def _predict(self, X, N, lsp, nls, m, C, f32_factor):
        dX = self.upload(X)
        if C is None:
            Ey = np.empty(N)
            try:
                if N:
                    _check(self.lib, self.lib.rr_rff_predict_mean_dev(self.h, dX.ptr, rr_dtype(dX.dtype), N, dX.ld, lsp, nls,
                                                                      m.ctypes.data_as(ctypes.c_void_p),
                                                                      Ey.ctypes.data_as(ctypes.c_void_p)))
            finally:
                dX.free()
            return Ey, None
        Ey, Vf = np.empty(N), np.empty(N)
        if N and f32_factor:
            B, form = C.factor()  # f32 arithmetic: variance as a sum of squares, no cancellation
            _check(self.lib, self.lib.rr_rff_predict_devb(self.h, dX.ptr, rr_dtype(dX.dtype), N, dX.ld, lsp, nls,
                                                          m.ctypes.data_as(ctypes.c_void_p), B.ptr, form,
                                                          Ey.ctypes.data_as(ctypes.c_void_p),
                                                          Vf.ctypes.data_as(ctypes.c_void_p)))
            dX.free()
            return Ey, Vf
        if isinstance(C, np.ndarray):
            C = np.ascontiguousarray(C, dtype=np.float64)
            fn, cp = self.lib.rr_rff_predict_dev, C.ctypes.data_as(ctypes.c_void_p)
        else:
            fn, cp = self.lib.rr_rff_predict_devc, _ptr(C)
        if N:
            _check(self.lib, fn(self.h, dX.ptr, rr_dtype(dX.dtype), N, dX.ld, lsp, nls,
                                                         m.ctypes.data_as(ctypes.c_void_p),
                                                         cp,
                                                         Ey.ctypes.data_as(ctypes.c_void_p),
                                                         Vf.ctypes.data_as(ctypes.c_void_p)))
        dX.free()
        return Ey, Vf

    def gram_timings(self):
        """(features_ms, syrk_ms, diag_ms, launches) of the last gram_dev call (waits for it)."""
        a, g, d, k = ctypes.c_float(), ctypes.c_float(), ctypes.c_float(), ctypes.c_int()
        _check(self.lib, self.lib.rr_rff_gram_timings(self.h, ctypes.byref(a), ctypes.byref(g), ctypes.byref(d),
                                                      ctypes.byref(k)))
        return a.value, g.value, d.value, k.value

    def symmetrize_dev(self, dG):
        _check(self.lib, self.lib.rr_symmetrize_dev(self.dev.ctx, _ptr(dG), 2 * self.n))

    def gram_kernel_name(self):
        return self.lib.rr_rff_gram_kernel_name(self.h).decode()
