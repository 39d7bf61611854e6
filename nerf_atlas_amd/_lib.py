"""ctypes loader for libnerf_atlas_amd.so (the C ABI of include/nerf_atlas_amd.h).

The HIP library is the product: there is no CPU or PyTorch fallback.  If the shared object is missing
(or a symbol of the header is missing from it) importing ops raises immediately.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# NA_LIB_PATH: another build of the same library (experiments such as tools/sched_fuzz.py full: the whole GPU suite against it)
LIB_PATH = os.environ.get("NA_LIB_PATH") or os.path.join(HERE, "libnerf_atlas_amd.so")

c_f32p = C.c_void_p  # device pointers travel as integers
c_i64 = C.c_int64


class NaMipDesc(C.Structure):
    """include/nerf_atlas_amd.h NaMipDesc: the crop an IPE latent is generated from inside the MLP prologue."""
    _fields_ = [("rays", C.c_void_p), ("ts", C.c_void_p), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("T", C.c_int32), ("kind", C.c_int32), ("min_deg", C.c_int32), ("max_deg", C.c_int32),
                ("t_end", C.c_float)]


class NaMlpDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "in_size", "enc_kind", "enc_dims", "latent_size", "num_layers", "hidden", "out_size", "skip",
        "activation", "layout")]


# name -> (restype, argtypes); mirrors include/nerf_atlas_amd.h one to one
SIGNATURES = {
    "na_version": (C.c_int, []),
    "na_last_error": (C.c_char_p, []),
    "na_raygen": (C.c_int, [c_f32p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p,
                            C.c_float, c_f32p, C.c_void_p]),
    "na_raygen_dtu": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p,
                                C.c_void_p]),
    "na_compute_ts": (C.c_int, [C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, c_f32p, c_f32p, c_f32p,
                                C.c_void_p]),
    "na_compute_pts": (C.c_int, [c_f32p, c_f32p, C.c_int, c_i64, c_f32p, C.c_void_p]),
    "na_hash_encode": (C.c_int, [c_f32p, c_i64, c_f32p, C.c_int, c_f32p, C.c_void_p, C.c_void_p]),
    "na_fourier_encode": (C.c_int, [c_f32p, c_i64, C.c_int, c_f32p, C.c_int, C.c_float, c_f32p, C.c_void_p]),
    "na_positional_encode": (C.c_int, [c_f32p, c_i64, C.c_int, c_f32p, C.c_int, c_f32p, C.c_void_p]),
    "na_view_elaz": (C.c_int, [c_f32p, c_i64, c_f32p, C.c_void_p]),
    "na_view_rows": (C.c_int, [c_f32p, c_f32p, c_i64, c_i64, c_f32p, C.c_void_p]),
    "na_hash_encode_rows": (C.c_int, [c_f32p, c_i64, c_f32p, C.c_int, C.c_int, c_f32p, C.c_void_p]),
    "na_plain_head_rows": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i64, c_i64, C.c_int, c_f32p, c_f32p, C.c_void_p]),
    "na_hash_encode_backward_rows": (C.c_int, [c_f32p, c_i64, c_f32p, C.c_int, C.c_int, c_f32p, C.c_void_p]),
    "na_hash_encode_backward_input_rows": (C.c_int, [c_f32p, c_i64, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, C.c_void_p]),
    "na_plain_head_rows_backward": (C.c_int, [c_f32p, c_f32p, c_i64, C.c_int, c_f32p, c_f32p, C.c_void_p]),
    "na_adam_step": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                               C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "na_sigmoid": (C.c_int, [c_f32p, c_i64, C.c_int, c_f32p, C.c_void_p]),
    "na_mip_encode": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, C.c_int, C.c_float, C.c_int,
                                C.c_int, c_f32p, C.c_void_p]),
    "na_composite": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, c_i64, C.c_int, C.c_int, C.c_int, c_f32p,
                               c_f32p, c_f32p, C.c_void_p]),
    "na_integrate": (C.c_int, [c_f32p, c_f32p, C.c_int, c_i64, C.c_int, c_f32p, C.c_void_p]),
    "na_composite_random_bg": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, c_i64, C.c_int, C.c_int, c_f32p, c_f32p,
                                         c_f32p, c_f32p, C.c_void_p]),
    "na_sky_random": (C.c_int, [c_f32p, c_f32p, C.c_int, c_i64, C.c_int, c_f32p, C.c_void_p]),
    "na_composite_random_bg_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, c_i64, C.c_int, C.c_int, c_f32p,
                                                  c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "na_normalize3": (C.c_int, [c_f32p, c_i64, c_f32p, C.c_void_p]),
    "na_pos_linear_combine": (C.c_int, [c_f32p, c_f32p, c_i64, c_i64, C.c_int, c_f32p, C.c_void_p]),
    "na_laplace_density": (C.c_int, [c_f32p, c_i64, c_f32p, c_f32p, C.c_void_p]),
    "na_bezier_warp": (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_i64, C.c_int, c_f32p, c_f32p, c_f32p,
                                 C.c_void_p]),
    "na_bezier_warp_latent": (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_i64, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p,
                                        c_f32p, C.c_void_p]),
    "na_linear_f32": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_i64, c_f32p, c_f32p, C.c_int, C.c_int, c_f32p,
                                C.c_void_p]),
    "na_mlp_packed_bytes": (C.c_size_t, [C.POINTER(NaMlpDesc), C.c_int]),
    "na_mlp_pack": (C.c_int, [C.POINTER(NaMlpDesc), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                              C.c_void_p, C.c_void_p]),
    "na_mlp_forward": (C.c_int, [C.POINTER(NaMlpDesc), C.c_int, C.c_void_p, c_f32p, c_f32p, c_f32p, c_i64, c_f32p,
                                 C.c_void_p]),
    "na_mlp_forward_ld": (C.c_int, [C.POINTER(NaMlpDesc), C.c_int, C.c_void_p, c_f32p, c_i64, c_f32p, c_i64, c_f32p,
                                    c_i64, c_f32p, C.c_void_p]),
    "na_mlp_forward_mip": (C.c_int, [C.POINTER(NaMlpDesc), C.c_int, C.c_void_p, c_f32p, c_i64, c_f32p, c_i64, c_f32p,
                                     C.c_void_p, c_i64, c_f32p, C.c_void_p]),
    "na_ray_points": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_float, c_i64, c_f32p, C.c_void_p]),
    "na_compact_rays": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "na_ray_points_indexed": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_void_p, c_i64, c_f32p, C.c_void_p]),
    "na_sphere_march_update_indexed": (C.c_int, [c_f32p, C.c_int, C.c_void_p, c_i64, C.c_float, C.c_float, c_f32p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]),
    "na_bisection_update_indexed": (C.c_int, [c_f32p, C.c_int, C.c_void_p, c_i64, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p,
                                              c_f32p, C.c_void_p, C.c_void_p]),
    "na_sphere_march_update": (C.c_int, [c_f32p, C.c_int, c_i64, C.c_float, C.c_float, c_f32p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "na_sign_change_update": (C.c_int, [c_f32p, C.c_int, c_i64, C.c_int, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "na_bisection_update": (C.c_int, [c_f32p, C.c_int, c_i64, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                      C.c_void_p, C.c_void_p]),
    "na_point_light": (C.c_int, [c_f32p, c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, c_i64, c_f32p, c_f32p, c_f32p,
                                 C.c_void_p]),
    "na_occlusion_apply": (C.c_int, [c_f32p, C.c_void_p, c_f32p, C.c_int, C.c_float, c_i64, c_f32p, C.c_void_p]),
    "na_set_deterministic": (C.c_int, [C.c_void_p, C.c_size_t]),
    "na_act_deriv": (C.c_int, [c_f32p, c_i64, C.c_int, C.c_int, c_f32p, C.c_void_p]),
    "na_mul_bcast": (C.c_int, [c_f32p, c_f32p, c_i64, C.c_int, c_f32p, C.c_void_p]),
    "na_mul_reduce": (C.c_int, [c_f32p, c_f32p, c_i64, C.c_int, c_f32p, C.c_void_p]),
    "na_eikonal_loss": (C.c_int, [c_f32p, c_i64, c_f32p, C.c_void_p]),
    "na_eikonal_loss_backward": (C.c_int, [c_f32p, c_i64, c_f32p, c_f32p, C.c_void_p]),
    "na_act_backward": (C.c_int, [c_f32p, c_f32p, c_i64, C.c_int, c_f32p, C.c_void_p]),
    "na_sigmoid_backward": (C.c_int, [c_f32p, c_f32p, c_i64, C.c_int, c_f32p, C.c_void_p]),
    "na_pos_linear_combine_backward": (C.c_int, [c_f32p, c_f32p, c_i64, c_f32p, c_i64, C.c_int, c_f32p, c_f32p, c_i64,
                                                 C.c_void_p]),
    "na_linear_bf16x3": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_i64, c_f32p, c_f32p, C.c_int, C.c_int, c_f32p,
                                   C.c_void_p]),
    "na_linear_dgrad_bf16x3": (C.c_int, [c_f32p, C.c_int, c_i64, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, c_f32p,
                                  c_f32p, C.c_void_p]),
    "na_linear_wgrad": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_i64, c_f32p, C.c_int, C.c_int, c_f32p, c_f32p,
                                  C.c_void_p]),
    "na_linear_wgrad_bf16x3": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_i64, c_f32p, C.c_int, C.c_int, c_f32p,
                                         c_f32p, C.c_void_p]),
    "na_linear_wgrad_bf16x3_ow": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_i64, c_f32p, C.c_int, C.c_int, c_f32p,
                                            c_f32p, C.c_void_p]),
    "na_train_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "na_train_gemm_packed_ok": (C.c_int, [c_i64, C.c_int]),
    "na_train_pack_many": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_void_p]),
    "na_linear_bf16x3_pk": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_i64, C.c_void_p, c_f32p, C.c_int, C.c_int, c_f32p,
                                      C.c_void_p]),
    "na_linear_dgrad_bf16x3_pk": (C.c_int, [c_f32p, C.c_int, c_i64, C.c_void_p, c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, c_f32p,
                                            c_f32p, C.c_void_p]),
    "na_linear_bwd_fused_ok": (C.c_int, [c_i64, C.c_int, C.c_int]),
    "na_train_packed_row_offset": (C.c_size_t, [C.c_int, C.c_int]),
    "na_linear_wgrad_bf16x3_cols": (C.c_int, [c_f32p, C.c_int, c_i64, c_f32p, C.c_int, C.c_int, c_f32p, C.c_int, c_f32p, C.c_void_p]),
    "na_linear_bwd_workspace_bytes": (C.c_size_t, [c_i64, C.c_int]),
    "na_linear_bwd_partial_count": (C.c_int, [c_i64, C.c_int]),
    "na_linear_bwd_partials_bf16x3_pk": (C.c_int, [c_f32p, C.c_int, c_i64, C.c_void_p, c_f32p, C.c_int, C.c_int, c_f32p, c_f32p, C.c_int,
                                                   C.c_void_p, C.c_void_p]),
    "na_train_reduce_many": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    "na_linear_bwd_bf16x3_pk": (C.c_int, [c_f32p, C.c_int, c_i64, C.c_void_p, c_f32p, C.c_int, C.c_int, c_f32p, c_f32p, C.c_int,
                                          c_f32p, C.c_void_p, C.c_void_p]),
    "na_hash_encode_backward": (C.c_int, [c_f32p, c_i64, c_f32p, C.c_int, c_f32p, C.c_void_p]),
    "na_hash_encode_backward_input": (C.c_int, [c_f32p, c_i64, c_f32p, c_f32p, C.c_int, c_f32p, C.c_void_p]),
    "na_hash_encode_jvp": (C.c_int, [c_f32p, c_i64, c_f32p, c_f32p, C.c_int, c_f32p, C.c_void_p]),
    "na_hash_encode_jvp_backward": (C.c_int, [c_f32p, c_f32p, c_i64, c_f32p, C.c_int, c_f32p, C.c_void_p]),
    "na_ffjord_div": (C.c_int, [c_f32p, c_f32p, C.c_int, c_f32p, c_f32p, c_i64, C.c_int, c_f32p, C.c_void_p]),
    "na_laplace_density_backward": (C.c_int, [c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "na_bezier_warp_backward": (C.c_int, [c_f32p, C.c_int, c_f32p, c_i64, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p,
                                          C.c_void_p]),
    "na_bezier_warp_latent_backward": (C.c_int, [c_f32p, C.c_int, c_f32p, c_i64, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p,
                                                 c_f32p, c_f32p, C.c_void_p]),
    "na_composite_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, c_i64, C.c_int, C.c_int, C.c_int,
                                        c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "na_render_workspace_bytes": (C.c_size_t, [C.c_int, c_i64]),
    "na_render_plain_view": (C.c_int, [c_f32p, c_i64, c_f32p, C.c_int, c_f32p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t,
                                       C.c_void_p]),
    "na_render_plain_view_pts": (C.c_int, [c_f32p, c_f32p, c_i64, c_f32p, C.c_int, c_f32p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t,
                                           C.c_void_p]),
    "na_mlp_fourier_ls_packed_bytes": (C.c_size_t, [C.c_int]),
    "na_mlp_fourier_ls_pack": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "na_mlp_fourier_ls": (C.c_int, [c_f32p, c_f32p, c_i64, c_f32p, C.c_int, c_f32p, C.c_void_p, C.c_int, c_f32p, c_i64, C.c_void_p]),
    "na_resample_ts_lds_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "na_resample_ts": (C.c_int, [c_f32p, c_f32p, c_i64, C.c_int, c_f32p, C.c_int, c_f32p, c_f32p, C.c_void_p]),
    "na_render_plain_view_ls_rayts": (C.c_int, [c_f32p, c_i64, c_f32p, C.c_int, c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "na_train_plain_view_ls": (C.c_int, [c_f32p, c_f32p, c_i64, c_f32p, C.c_int, c_f32p, C.c_void_p, C.c_int, c_f32p, c_f32p,
                                         c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "na_render_plain_mip_ls_packed_bytes": (C.c_size_t, [C.c_int]),
    "na_render_plain_mip_ls_pack": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "na_render_plain_mip_ls": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, c_f32p, C.c_void_p, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                         C.c_size_t, C.c_void_p]),
    "na_render_plain_pos_ls_packed_bytes": (C.c_size_t, [C.c_int]),
    "na_render_plain_plv_ls_packed_bytes": (C.c_size_t, [C.c_int]),
    "na_render_head_ls_workspace_bytes": (C.c_size_t, [C.c_int, c_i64]),
    "na_render_plain_pos_ls_pack": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "na_render_plain_plv_ls_pack": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p]),
    "na_render_plain_pos_ls": (C.c_int, [c_f32p, c_f32p, c_i64, c_f32p, C.c_int, c_f32p, c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "na_render_plain_plv_ls": (C.c_int, [c_f32p, c_f32p, c_i64, c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, c_i64, C.c_int, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "na_mlp_hash_ls_packed_bytes": (C.c_size_t, [C.c_int]),
    "na_mlp_hash_ls_pack": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "na_mlp_hash_ls": (C.c_int, [c_f32p, c_f32p, c_i64, c_f32p, C.c_int, c_f32p, C.c_void_p, C.c_int, C.c_int, c_f32p, c_i64,
                                 C.c_void_p]),
    "na_render_tiny_ls_packed_bytes": (C.c_size_t, [C.c_int]),
    "na_render_tiny_ls_pack": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "na_render_tiny_ls": (C.c_int, [c_f32p, c_f32p, c_i64, c_f32p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, c_f32p,
                                    c_f32p, c_f32p, C.c_void_p]),
    "na_render_view_ls_packed_bytes": (C.c_size_t, [C.c_int]),
    "na_render_view_ls_pack": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "na_render_view_ls": (C.c_int, [c_f32p, c_f32p, c_i64, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "na_render_volsdf_siren_ls_packed_bytes": (C.c_size_t, [C.c_int]),
    "na_render_volsdf_siren_ls_pack": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                 C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "na_render_volsdf_siren_ls": (C.c_int, [c_f32p, c_f32p, c_i64, c_f32p, C.c_int, c_f32p, C.c_void_p, C.c_int, C.c_int,
                                            C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "na_render_ls_packed_bytes": (C.c_size_t, [C.c_int]),
    "na_render_ls_pack": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "na_render_ls_workspace_bytes": (C.c_size_t, [C.c_int, c_i64]),
    "na_render_plain_view_ls": (C.c_int, [c_f32p, c_f32p, c_i64, c_f32p, C.c_int, c_f32p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
}

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def load():
    """Load the HIP library (after torch, so both share one libamdhip64)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: the HIP extension is the only implementation of the hot path. "
            "Build it with `python -m nerf_atlas_amd.build` (or __graft_entry__.build()).")
    import torch  # noqa: F401  (loads libamdhip64 first)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryMissing(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class NaError(RuntimeError):
    def __init__(self, code, msg=None):
        if msg is None:  # raised by the host layer itself (ops.py): NaError("...") -- no C status code
            code, msg = -1, code
        super().__init__(f"nerf_atlas_amd error {code}: {msg}")
        self.code = code


def check(code):
    if code != 0:
        msg = load().na_last_error()
        raise NaError(code, msg.decode() if msg else "")
