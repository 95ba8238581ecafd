"""ctypes declarations for include/kaldi_hip.h (libkaldi_hip.so).

The library is the product; this module only declares its C-ABI.  Loading fails
loudly if the shared object is missing — there is no fallback path.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KH_LIB_OVERRIDE") or os.path.join(HERE, "libkaldi_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_double_p = C.POINTER(C.c_double)
vp = C.c_void_p


class KhMatrixDim(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("stride", C.c_int32)]


class KhComponentDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("input_dim", C.c_int32), ("output_dim", C.c_int32),
        ("linear", c_float_p), ("bias", c_float_p),
        ("context", c_int32_p), ("n_context", C.c_int32), ("const_dim", C.c_int32),
        ("p", C.c_float), ("sizes", c_int32_p), ("n_sizes", C.c_int32),
    ]


class KhDecoderConfig(C.Structure):
    _fields_ = [
        ("beam", C.c_float), ("max_active", C.c_int32), ("min_active", C.c_int32),
        ("lattice_beam", C.c_float), ("prune_interval", C.c_int32),
        ("beam_delta", C.c_float), ("hash_ratio", C.c_float), ("prune_scale", C.c_float),
    ]


class KhMfccOptions(C.Structure):
    _fields_ = [("snip_edges", C.c_int32), ("use_energy", C.c_int32), ("raw_energy", C.c_int32), ("htk_compat", C.c_int32),
                ("energy_floor", C.c_float), ("dither", C.c_float), ("dither_seed", C.c_uint64)]


class KhIvectorConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "base_dim", "splice_left", "splice_right", "feat_dim", "num_gauss", "ivector_dim", "lda_cols",
        "cmn_window", "speaker_frames", "global_frames", "normalize_mean", "normalize_variance",
        "ivector_period", "num_gselect", "num_cg_iters")] + [
        ("min_post", C.c_float), ("posterior_scale", C.c_float), ("max_count", C.c_float),
        ("prior_offset", C.c_double), ("greedy_most_recent", C.c_int32)]


class KhDecodeStats(C.Structure):
    _fields_ = [
        ("num_frames", C.c_int32), ("reached_final", C.c_int32),
        ("final_relative_cost", C.c_float), ("final_best_cost", C.c_float),
        ("num_tokens", C.c_int32), ("num_links", C.c_int32),
        ("arcs_expanded", C.c_int64), ("tokens_created", C.c_int64),
        ("status", C.c_int32), ("max_tokens_frame", C.c_int32),
    ]


D = KhMatrixDim
f, i32, i64 = C.c_float, C.c_int32, C.c_int64

# name -> (restype, argtypes).  Must list EVERY symbol include/kaldi_hip.h declares
# (tests/test_abi.py checks this against the header).
SIGNATURES = {
    "kh_last_error": (C.c_char_p, []),
    "kh_device_count": (C.c_int, []),
    "kh_select_gpu": (C.c_int, [C.c_int]),
    "kh_enabled": (C.c_int, []),
    "kh_device_name": (C.c_int, [C.c_char_p, C.c_size_t]),
    "kh_mem_info": (C.c_int, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "kh_set_stream": (C.c_int, [vp]),
    "kh_get_stream": (vp, []),
    "kh_synchronize": (C.c_int, []),
    "kh_malloc": (vp, [C.c_size_t]),
    "kh_malloc_pitch": (vp, [C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]),
    "kh_free": (C.c_int, [vp]),
    "kh_pool_release": (C.c_int, []),
    "kh_memcpy_2d": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]),
    "kh_memset": (C.c_int, [vp, C.c_int, C.c_size_t]),
    "kh_add_mat_mat": (C.c_int, [f, vp, D, C.c_int, vp, D, C.c_int, f, vp, D]),
    "kh_affine": (C.c_int, [vp, D, vp, D, vp, vp, D]),
    "kh_affine_pnorm": (C.c_int, [vp, D, vp, D, vp, vp, D, C.c_int]),
    "kh_affine_pnorm_supported": (C.c_int, [C.c_int]),
    "kh_softmax_per_row": (C.c_int, [vp, vp, D, C.c_int]),
    "kh_log_softmax_per_row": (C.c_int, [vp, vp, D, C.c_int]),
    "kh_copy_rows": (C.c_int, [vp, D, vp, C.c_int, vp]),
    "kh_splice": (C.c_int, [vp, D, vp, D, vp, C.c_int]),
    "kh_group_pnorm": (C.c_int, [vp, vp, D, C.c_int, C.c_int, f]),
    "kh_normalize": (C.c_int, [vp, vp, D, C.c_int]),
    "kh_add_diag_mat2": (C.c_int, [f, vp, D, f, vp]),
    "kh_mul_rows_vec": (C.c_int, [vp, D, vp]),
    "kh_mul_cols_vec": (C.c_int, [vp, D, vp]),
    "kh_copy_rows_from_vec": (C.c_int, [vp, D, vp]),
    "kh_add_vec_to_rows": (C.c_int, [f, vp, f, vp, D]),
    "kh_apply_floor": (C.c_int, [vp, D, f]),
    "kh_apply_log": (C.c_int, [vp, D]),
    "kh_apply_exp": (C.c_int, [vp, D]),
    "kh_apply_pow": (C.c_int, [vp, D, f]),
    "kh_scale": (C.c_int, [vp, D, f]),
    "kh_sum_column_ranges": (C.c_int, [vp, D, vp, D, vp]),
    "kh_matrix_lookup": (C.c_int, [vp, D, vp, C.c_int, vp]),
    "kh_log_prior_scale": (C.c_int, [vp, D, vp, f]),
    "kh_nnet_create": (vp, []),
    "kh_nnet_destroy": (None, [vp]),
    "kh_nnet_add_component": (C.c_int, [vp, C.POINTER(KhComponentDesc)]),
    "kh_nnet_set_priors": (C.c_int, [vp, c_float_p, C.c_int]),
    "kh_nnet_num_components": (C.c_int, [vp]),
    "kh_nnet_input_dim": (C.c_int, [vp]),
    "kh_nnet_output_dim": (C.c_int, [vp]),
    "kh_nnet_left_context": (C.c_int, [vp]),
    "kh_nnet_right_context": (C.c_int, [vp]),
    "kh_nnet_compute": (C.c_int, [vp, vp, C.c_int, c_int32_p, C.c_int, C.c_int, C.c_int, f, vp, C.c_int, c_int32_p]),
    "kh_nnet_compute_async": (C.c_int, [vp, vp, C.c_int, c_int32_p, C.c_int, C.c_int, C.c_int, f, vp, C.c_int, c_int32_p]),
    "kh_gmm_compute_gconsts": (C.c_int, [c_float_p, c_float_p, c_float_p, C.c_int, C.c_int, c_float_p]),
    "kh_diag_gmm_loglikes": (C.c_int, [vp, D, vp, vp, vp, C.c_int, vp, C.c_int]),
    "kh_am_gmm_loglikes": (C.c_int, [vp, D, vp, vp, vp, vp, C.c_int, C.c_int, f, vp, C.c_int]),
    "kh_fst_create": (vp, [i32, i32, c_int64_p, c_int32_p, c_int32_p, c_float_p, c_int32_p, c_float_p]),
    "kh_fst_destroy": (None, [vp]),
    "kh_fst_num_arcs": (i64, [vp]),
    "kh_fst_check_pdf_map": (C.c_int, [vp, c_int32_p, C.c_int, C.c_int]),
    "kh_decoder_config_default": (None, [C.POINTER(KhDecoderConfig)]),
    "kh_decoder_create": (vp, [vp, C.POINTER(KhDecoderConfig), C.c_int, C.c_int]),
    "kh_decoder_destroy": (None, [vp]),
    "kh_decoder_decode": (C.c_int, [vp, vp, C.c_int, c_int32_p, C.c_int, vp]),
    "kh_add_mat_mat_d": (C.c_int, [C.c_double, vp, D, C.c_int, vp, D, C.c_int, C.c_double, vp, D]),
    "kh_softmax_per_row_d": (C.c_int, [vp, vp, D, C.c_int]),
    "kh_log_softmax_per_row_d": (C.c_int, [vp, vp, D, C.c_int]),
    "kh_copy_rows_d": (C.c_int, [vp, D, vp, C.c_int, vp]),
    "kh_splice_d": (C.c_int, [vp, D, vp, D, vp, C.c_int]),
    "kh_group_pnorm_d": (C.c_int, [vp, vp, D, C.c_int, C.c_int, C.c_double]),
    "kh_add_diag_mat2_d": (C.c_int, [C.c_double, vp, D, C.c_double, vp]),
    "kh_mul_rows_vec_d": (C.c_int, [vp, D, vp]),
    "kh_mul_cols_vec_d": (C.c_int, [vp, D, vp]),
    "kh_copy_rows_from_vec_d": (C.c_int, [vp, D, vp]),
    "kh_add_vec_to_rows_d": (C.c_int, [C.c_double, vp, C.c_double, vp, D]),
    "kh_apply_floor_d": (C.c_int, [vp, D, C.c_double]),
    "kh_apply_log_d": (C.c_int, [vp, D]),
    "kh_apply_exp_d": (C.c_int, [vp, D]),
    "kh_apply_pow_d": (C.c_int, [vp, D, C.c_double]),
    "kh_scale_d": (C.c_int, [vp, D, C.c_double]),
    "kh_sum_column_ranges_d": (C.c_int, [vp, D, vp, D, vp]),
    "kh_matrix_lookup_d": (C.c_int, [vp, D, vp, C.c_int, vp]),
    "kh_online_decoder_set_pdf_map": (C.c_int, [vp, vp, C.c_int]),
    "kh_online_nnet2_create": (vp, [vp, vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]),
    "kh_online_nnet2_destroy": (None, [vp]),
    "kh_online_nnet2_reset": (C.c_int, [vp, vp, C.c_int]),
    "kh_online_nnet2_step": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp]),
    "kh_online_nnet2_num_frames_ready": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int32)]),
    "kh_online_nnet2_serve_start": (C.c_int, [vp, vp]),
    "kh_online_nnet2_serve_stop": (C.c_int, [vp]),
    "kh_online_nnet2_serve_finalize": (C.c_int, [vp, C.POINTER(C.c_int32), C.c_int]),
    "kh_online_nnet2_serve_poll": (C.c_int, [vp, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "kh_online_nnet2_serve_wait": (C.c_int, [vp, C.POINTER(C.c_int32), C.c_int, C.c_int]),
    "kh_online_decoder_serve_start": (C.c_int, [vp, vp, C.c_int, C.c_int64, vp]),
    "kh_online_decoder_serve_stop": (C.c_int, [vp]),
    "kh_online_decoder_set_lazy_prune": (C.c_int, [vp, C.c_int]),
    "kh_online_decoder_set_reference_order": (C.c_int, [vp, C.c_int]),
    "kh_online_decoder_serve_init": (C.c_int, [vp, C.POINTER(C.c_int32), C.c_int]),
    "kh_online_decoder_serve_publish": (C.c_int, [vp, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32)]),
    "kh_online_decoder_serve_finalize": (C.c_int, [vp, C.POINTER(C.c_int32), C.c_int]),
    "kh_online_decoder_serve_poll": (C.c_int, [vp, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "kh_online_decoder_serve_wait": (C.c_int, [vp, C.POINTER(C.c_int32), C.c_int, C.c_int]),
    "kh_lattice_batch_create": (vp, [C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "kh_lattice_batch_destroy": (None, [vp]),
    "kh_lattice_batch_sizes": (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "kh_lattice_batch_forward_backward": (C.c_int, [vp, vp, vp, vp, vp]),
    "kh_lattice_batch_forward_backward_dev": (C.c_int, [vp, vp, c_double_p, c_double_p]),
    "kh_lattice_batch_rescore": (C.c_int, [vp, vp, C.c_int, vp, vp, vp]),
    "kh_lattice_last_timings": (C.c_int, [C.POINTER(C.c_float)]),
    "kh_decoder_set_reference_order": (C.c_int, [vp, C.c_int]),
    "kh_decoder_get_search_counters": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int64)]),
    "kh_decoder_get_stats": (C.c_int, [vp, C.c_int, C.POINTER(KhDecodeStats)]),
    "kh_decoder_get_counters": (C.c_int, [vp, C.c_int, C.POINTER(KhDecodeStats)]),
    "kh_decoder_get_schedule_counters": (C.c_int, [vp, C.c_int, c_int32_p]),
    "kh_decoder_last_kernel_ms": (C.c_int, [vp, c_float_p]),
    "kh_decoder_last_host_tail_ms": (C.c_int, [vp, c_float_p]),
    "kh_decoder_set_after_launch": (C.c_int, [vp, vp, vp]),
    "kh_decoder_get_raw_lattice": (C.c_int, [vp, C.c_int, c_int32_p, c_int32_p, c_float_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_float_p, c_float_p]),
    "kh_decoder_get_best_path": (C.c_int, [vp, C.c_int, c_int32_p, C.c_int, c_int32_p, c_int32_p, C.c_int, c_int32_p, c_float_p, c_float_p]),
    "kh_decoder_get_best_paths": (C.c_int, [vp, C.c_int, C.c_int, c_int32_p, C.c_int64, C.POINTER(C.c_int64), c_int32_p, C.c_int64,
                                            C.POINTER(C.c_int64), c_float_p, c_float_p]),
    "kh_decoder_get_stats_batch": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(KhDecodeStats), C.POINTER(KhDecodeStats)]),
    "kh_decoder_prepare": (C.c_int, [vp, C.c_int]),
    "kh_decoder_set_determinize": (C.c_int, [vp, C.c_int, C.c_double, C.c_float, C.c_int64, c_int32_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "kh_decoder_get_compact_lattice": (vp, [vp, C.c_int]),
    "kh_decoder_compact_lattice_totals": (C.c_int, [vp, C.POINTER(C.c_int64)]),
    "kh_online_decoder_create": (vp, [vp, C.POINTER(KhDecoderConfig), C.c_int, C.c_int]),
    "kh_online_decoder_destroy": (None, [vp]),
    "kh_online_decoder_init_decoding": (C.c_int, [vp, c_int32_p, C.c_int]),
    "kh_online_decoder_advance": (C.c_int, [vp, c_int32_p, C.c_int, C.POINTER(vp), C.c_int, c_int32_p, vp]),
    "kh_online_decoder_num_frames_decoded": (C.c_int, [vp, C.c_int, c_int32_p]),
    "kh_online_decoder_finalize": (C.c_int, [vp, c_int32_p, C.c_int]),
    "kh_online_decoder_get_stats": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(KhDecodeStats)]),
    "kh_online_decoder_get_raw_lattice": (C.c_int, [vp, C.c_int, C.c_int, c_int32_p, c_int32_p, c_float_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_float_p, c_float_p]),
    "kh_online_decoder_get_best_path": (C.c_int, [vp, C.c_int, C.c_int, c_int32_p, C.c_int, c_int32_p, c_int32_p, C.c_int, c_int32_p, c_float_p, c_float_p]),
    "kh_mfcc_compute": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, c_float_p, C.c_int, c_int32_p, c_int32_p, c_float_p, C.c_int, c_float_p, c_float_p, vp, C.c_int, c_int32_p]),
    "kh_mfcc_compute_opts": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, c_float_p, C.c_int, c_int32_p, c_int32_p, c_float_p, C.c_int, c_float_p, c_float_p, C.POINTER(KhMfccOptions), vp, C.c_int, c_int32_p]),
    "kh_compute_deltas": (C.c_int, [vp, KhMatrixDim, C.c_int, c_float_p, c_int32_p, vp, C.c_int]),
    "kh_acc_cmvn_stats": (C.c_int, [vp, KhMatrixDim, c_double_p]),
    "kh_determinize_lattice_pruned": (vp, [C.c_int, C.c_int, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_float_p, c_float_p, c_float_p, C.c_double, C.c_float, C.c_int64]),
    "kh_determinize_lattice_phone_pruned": (vp, [C.c_int, C.c_int, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_float_p, c_float_p, c_float_p,
                                                 c_int32_p, C.c_int, C.c_double, C.c_float, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "kh_compact_lattice_sizes": (C.c_int, [vp, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    "kh_compact_lattice_get": (C.c_int, [vp, c_int32_p, c_int32_p, c_int32_p, c_float_p, c_float_p, c_int32_p, c_int32_p, c_float_p, c_float_p, c_int32_p, c_int32_p]),
    "kh_compact_lattice_free": (None, [vp]),
    "kh_ivector_extractor_create": (vp, [C.POINTER(KhIvectorConfig), c_float_p, c_double_p, c_float_p, c_float_p, c_float_p, c_double_p, c_double_p]),
    "kh_ivector_extractor_destroy": (None, [vp]),
    "kh_ivector_extract": (C.c_int, [vp, vp, C.c_int, c_int32_p, C.c_int, vp, C.c_int]),
    "kh_ivector_state_dim": (C.c_int, [vp]),
    "kh_ivector_extract_adapt": (C.c_int, [vp, vp, C.c_int, c_int32_p, C.c_int, c_double_p, c_double_p, vp, C.c_int]),
    "kh_ivector_streams_create": (vp, [vp, vp, C.c_int, c_int32_p, C.c_int, c_double_p, vp, C.c_int]),
    "kh_ivector_streams_destroy": (None, [vp]),
    "kh_ivector_streams_update_frame_weights": (C.c_int, [vp, C.c_int, C.c_int, c_int32_p, c_float_p, C.c_int]),
    "kh_ivector_streams_get_frames": (C.c_int, [vp, C.c_int, c_int32_p, c_int32_p]),
    "kh_ivector_streams_get_stats": (C.c_int, [vp, c_double_p]),
    "kh_lattice_state_times": (C.c_int, [C.c_int, c_int32_p, c_int64_p, c_int32_p, c_int32_p, c_float_p, c_int32_p, c_int32_p]),
    "kh_lattice_forward_backward": (C.c_int, [C.c_int, c_int32_p, c_int64_p, c_int32_p, c_int32_p, c_float_p, c_float_p, c_float_p, c_float_p, c_double_p, c_double_p, c_int32_p]),
    "kh_lattice_alphas_betas": (C.c_int, [C.c_int, c_int32_p, c_int64_p, c_int32_p, c_int32_p, c_float_p, c_float_p, c_float_p, C.c_int, c_double_p, c_double_p, c_double_p]),
    "kh_lattice_forward_backward_mpe": (C.c_int, [C.c_int, c_int32_p, c_int64_p, c_int32_p, c_int32_p, c_float_p, c_float_p, c_float_p, c_int32_p, c_int32_p, C.c_int, c_int32_p, C.c_int, c_int32_p, c_int32_p, C.c_int, C.c_int, c_float_p, c_double_p, c_int32_p]),
    "kh_rescore_lattice": (C.c_int, [C.c_int, c_int32_p, c_int64_p, c_int32_p, c_int32_p, c_float_p, vp, C.c_int, c_int32_p, vp]),
    "kh_merge_pair_vector_summing": (C.c_int, [i64, c_int32_p, c_int32_p, c_float_p, i32, c_int32_p, c_int32_p, c_float_p, c_int64_p]),
    "kh_discriminative_lattice_computations": (C.c_int, [C.c_int, c_int32_p, c_int64_p, c_int32_p, c_int32_p, c_float_p, c_float_p, c_float_p,
                                                        c_int32_p, c_int32_p, c_float_p, c_int32_p, c_int32_p, C.c_int, c_int32_p, C.c_int,
                                                        C.c_int, C.c_float, C.c_int, C.c_int, c_float_p, vp, KhMatrixDim, vp, KhMatrixDim, c_double_p]),
    "kh_discriminative_lattice_computations_parts": (C.c_int, [C.c_int, c_int32_p, vp, vp, vp, vp, vp, vp,
                                                              c_int32_p, c_int32_p, c_float_p, c_int32_p, c_int32_p, C.c_int, c_int32_p, C.c_int,
                                                              C.c_int, C.c_float, C.c_int, C.c_int, c_float_p, vp, KhMatrixDim, vp, KhMatrixDim, c_double_p]),
    "kh_discriminative_lattice_computations_begin": (C.c_int, [C.c_int, c_int32_p, vp, vp, vp, vp, vp, vp,
                                                              c_int32_p, c_int32_p, c_float_p, c_int32_p, c_int32_p, C.c_int, c_int32_p, C.c_int,
                                                              C.c_int, C.c_float, C.c_int, C.c_int, c_float_p, vp, KhMatrixDim, vp, KhMatrixDim, C.POINTER(vp)]),
    "kh_discriminative_lattice_computations_end": (C.c_int, [vp, c_double_p]),
    "kh_comp_objf_and_deriv": (C.c_int, [C.c_int, c_int32_p, c_int32_p, c_float_p, vp, KhMatrixDim, vp, KhMatrixDim, c_float_p, c_float_p]),
}

_lib = None


def load():
    """Load libkaldi_hip.so and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libkaldi_hip.so not found at %s — build it with `python __graft_entry__.py` "
            "(there is no CPU fallback)" % LIB_PATH)
    # One HIP runtime per process: the torch wheel carries its own libamdhip64 (same SONAME as /opt/rocm's).  Whichever
    # is loaded first serves both torch and this library; if libkaldi_hip.so came first and torch second, each would
    # initialise its own copy and the second one finds no device (seen as "no ROCm-capable device" in smoke() after
    # build()).  The Python layers above hold their device buffers in torch tensors, so torch's copy goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("KH_LIB_OVERRIDE") and not hasattr(lib, name):
            continue  # A/B runs against an older build of the library (tools/): it may predate a symbol
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class KhError(RuntimeError):
    """KALDI_ERR equivalent (base/kaldi-error.cc:143,179-182 throws std::runtime_error)."""


def check(rc):
    if rc != 0:
        raise KhError("libkaldi_hip error %d: %s" % (rc, load().kh_last_error().decode()))
