"""ctypes binding of liboess.so (the C-ABI declared in include/oess.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent the
import of any compute entry point raises.  Build it with ``python -c "import __graft_entry__ as g;
g.build()"`` or ``make -C openess_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OESS_LIB_PATH") or os.path.join(_HERE, "liboess.so")      # override: A/B builds of the same ABI

ABI_VERSION = 10         # == OESS_ABI_VERSION of include/oess.h (tests/test_abi.py keeps the two equal)

c_i64 = ctypes.c_int64
c_ll = ctypes.c_longlong
c_int = ctypes.c_int
c_f = ctypes.c_float
c_d = ctypes.c_double
c_vp = ctypes.c_void_p
c_sz = ctypes.c_size_t
c_u64 = ctypes.c_uint64

class ConvLstmDesc(ctypes.Structure):
    """oess_convlstm_desc_t (include/oess.h): one problem of oess_convlstm_fused_group_bf16."""
    _fields_ = [("in_", c_vp), ("in_pix_stride", c_ll), ("B", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int),
                ("w_packed_gates", c_vp), ("bias", c_vp), ("C_hidden", c_int), ("R", c_int), ("S", c_int), ("pad", c_int),
                ("prev_cell", c_vp), ("cell", c_vp), ("hidden", c_vp), ("hidden_pix_stride", c_ll)]


class ConvS2Desc(ctypes.Structure):
    """oess_conv_s2_desc_t (include/oess.h): one problem of oess_conv5x5s2_group_bf16."""
    _fields_ = [("in_", c_vp), ("in_pix_stride", c_ll), ("B", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int),
                ("w_packed", c_vp), ("bias", c_vp), ("Cout", c_int), ("relu", c_int), ("out", c_vp), ("out_pix_stride", c_ll)]


# name -> (restype, argtypes).  Must list EVERY symbol of include/oess.h (tests/test_abi.py checks).
SIGNATURES = {
    "oess_abi_version": (c_int, []),
    "oess_build_info": (ctypes.c_char_p, []),
    "oess_strerror": (ctypes.c_char_p, [c_int]),
    "oess_voxelize_workspace_bytes": (c_sz, [c_i64, c_int, c_i64, c_int, c_int, c_int, c_int]),
    "oess_voxelize_trilinear_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_int, c_int, c_int,
                                            c_int, c_vp, c_vp, c_sz, c_vp]),
    "oess_voxelize_dsec_raw": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_i64, c_int, c_int,
                                       c_int, c_int, c_int, c_vp, c_vp, c_sz, c_vp]),
    "oess_voxelize_nearest_i64": (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_vp,
                                          c_vp, c_sz, c_vp]),
    "oess_voxelize_nearest_f64": (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_vp,
                                          c_vp, c_sz, c_vp]),
    "oess_event_histogram_i64": (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_int, c_vp, c_vp]),
    "oess_masked_stats_doubles": (c_sz, [c_int]),
    "oess_masked_normalize_f32": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "oess_masked_normalize_slice_f32": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_i64, c_vp, c_vp]),
    "oess_segment_mean_fwd_workspace_bytes": (c_sz, [c_int, c_int]),
    "oess_segment_mean_fwd": (c_int, [c_vp, c_int, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "oess_segment_mean_bwd": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_int, c_vp, c_sz, c_vp]),
    "oess_task_loss_sums_doubles": (c_sz, [c_int]),
    "oess_task_loss_fwd": (c_int, [c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp,
                                   c_vp, c_vp]),
    "oess_task_loss_bwd": (c_int, [c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp,
                                   c_f, c_vp, c_vp, c_int, c_vp]),
    "oess_confusion_accumulate": (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp]),
    "oess_convlstm_gates_bf16": (c_int, [c_vp, c_ll, c_vp, c_vp, c_vp, c_ll, c_ll, c_int, c_vp]),
    "oess_convlstm_fused_bf16": (c_int, [c_vp, c_ll, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp,
                                         c_vp, c_vp, c_ll, c_vp]),
    "oess_convlstm_fused_group_bf16": (c_int, [c_vp, c_int, c_vp]),
    "oess_convlstm_w128_cell_bytes": (c_sz, [c_ll, c_int]),
    "oess_convlstm_w128_cell_relayout": (c_int, [c_vp, c_vp, c_ll, c_int, c_int, c_vp]),
    "oess_convlstm_w128_group_bf16": (c_int, [c_vp, c_int, c_vp]),
    "oess_convlstm_w128_tile_lists": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    "oess_conv5x5s2_group_bf16": (c_int, [c_vp, c_int, c_vp]),
    "oess_loss_partials_bytes": (c_sz, []),
    "oess_l1_mean_fwd": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "oess_l1_mean_bwd": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp]),
    "oess_cosine_mean_fwd": (c_int, [c_vp, c_ll, c_vp, c_ll, c_i64, c_int, c_int, c_f, c_vp, c_vp, c_vp]),
    "oess_cosine_mean_bwd": (c_int, [c_vp, c_ll, c_vp, c_ll, c_i64, c_int, c_int, c_f, c_vp, c_vp, c_ll, c_vp, c_ll, c_vp]),
    "oess_resize_bilinear_nhwc_fwd": (c_int, [c_vp, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_ll, c_vp]),
    "oess_resize_bilinear_bwd_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "oess_resize_bilinear_nhwc_bwd": (c_int, [c_vp, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_sz,
                                              c_vp, c_ll, c_vp]),
    "oess_l2norm_nhwc_fwd": (c_int, [c_vp, c_ll, c_i64, c_int, c_int, c_f, c_vp, c_ll, c_vp, c_vp]),
    "oess_l2norm_nhwc_bwd": (c_int, [c_vp, c_ll, c_vp, c_ll, c_vp, c_i64, c_int, c_int, c_f, c_vp, c_ll, c_vp]),
    "oess_zero_insert_nhwc_bf16": (c_int, [c_vp, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_ll, c_vp]),
    "oess_batchnorm_bwd_nhwc_bf16": (c_int, [c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_vp, c_int, c_ll, c_int, c_vp, c_vp,
                                             c_vp, c_ll, c_vp, c_ll, c_vp, c_sz, c_vp]),
    "oess_layernorm_bf16": (c_int, [c_vp, c_ll, c_i64, c_int, c_vp, c_vp, c_f, c_vp, c_ll, c_vp]),
    "oess_attention_d64_bf16": (c_int, [c_vp, c_ll, c_int, c_int, c_int, c_f, c_vp, c_ll, c_vp]),
    "oess_nce_loss_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_f, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "oess_nce_loss_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "oess_adamw_multi_f32": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_d, c_d, c_d, c_d, c_d, c_d, c_d, c_vp]),
    "oess_masked_stats_slice_f32": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_i64, c_vp, c_vp]),
    "oess_masked_stats_slices_f32": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_i64, c_vp, c_vp]),
    "oess_event_slice_to_nhwc8_bf16": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_ll, c_vp, c_int, c_vp, c_vp]),
    "oess_norm_partials_bytes": (c_sz, [c_int, c_ll, c_int, c_int]),
    "oess_norm_stats_nhwc_bf16": (c_int, [c_vp, c_ll, c_int, c_ll, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "oess_norm_stats_finalize_nhwc_bf16": (c_int, [c_vp, c_ll, c_int, c_ll, c_int, c_f, c_vp, c_vp, c_vp, c_vp, c_f, c_vp, c_vp,
                                                   c_vp, c_vp, c_vp, c_sz, c_vp]),
    "oess_norm_apply_nhwc_bf16": (c_int, [c_vp, c_ll, c_vp, c_vp, c_vp, c_ll, c_int, c_int, c_ll, c_int, c_vp, c_ll, c_vp]),
    "oess_instnorm_bwd_nhwc_bf16": (c_int, [c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_int, c_int, c_ll, c_int, c_vp, c_vp, c_vp,
                                            c_ll, c_vp, c_sz, c_vp]),
    "oess_upsample_nearest2x_nhwc_bf16": (c_int, [c_vp, c_ll, c_int, c_int, c_int, c_int, c_vp, c_ll, c_vp]),
    "oess_downsample_sum2x_nhwc_bf16": (c_int, [c_vp, c_ll, c_int, c_int, c_int, c_int, c_vp, c_ll, c_vp]),
    "oess_bilinear_l2norm_nhwc_bf16": (c_int, [c_vp, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_ll, c_vp, c_vp]),
    "oess_conv2d_wgrad_bf16": (c_int, [c_vp, c_ll, c_int, c_int, c_int, c_int, c_vp, c_ll, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_vp, c_vp, c_sz, c_vp]),
    "oess_bilinear_l2norm_pool_bwd_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int, c_int]),
    "oess_bilinear_l2norm_pool_bwd_bf16": (c_int, [c_vp, c_ll, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                   c_f, c_vp, c_sz, c_vp, c_ll, c_vp]),
    "oess_pool_matrix_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "oess_pool_matrix_build": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp]),
    "oess_pool_matrix_fwd": (c_int, [c_vp, c_vp, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "oess_pool_matrix_bwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_ll, c_int, c_vp]),
    "oess_linear_probe_partials_bytes": (c_sz, [c_int]),
    "oess_linear_probe_fwd_f32": (c_int, [c_vp, c_vp, c_vp, c_ll, c_int, c_vp, c_vp]),
    "oess_linear_probe_bwd_f32": (c_int, [c_vp, c_vp, c_vp, c_ll, c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "oess_conv2d_pack_multi_blocks": (c_ll, [c_int, c_int, c_int, c_int, c_int]),
    "oess_conv2d_pack_weight_multi": (c_int, [c_vp, c_int, c_ll, c_int, c_vp]),
    "oess_conv2d_packed_bytes": (c_sz, [c_int, c_int, c_int, c_int, c_int]),
    "oess_conv2d_pack_weight": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp]),
    "oess_e2vid_events_head_enc0_bf16": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_ll, c_vp]),
    "oess_e2vid_head_enc0_bf16": (c_int, [c_vp, c_ll, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_ll, c_vp]),
    "oess_conv2d_fwd_bf16": (c_int, [c_vp, c_ll, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_vp, c_ll, c_vp, c_vp, c_ll, c_vp, c_vp, c_sz, c_vp]),
    "oess_conv2d_fwd_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "oess_norm_tile_stats_apply_nhwc_bf16": (c_int, [c_vp, c_int, c_int, c_f, c_f, c_vp, c_vp, c_vp, c_vp, c_f, c_vp, c_vp, c_vp, c_ll,
                                                     c_vp, c_ll, c_int, c_ll, c_vp, c_ll, c_vp]),
    "oess_png_decode_scratch_bytes": (c_sz, [c_ll, c_int, c_int, c_int]),
    "oess_png_decode_gray8_batch": (c_int, [c_vp, c_vp, c_ll, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "oess_norm_reduce_finalize_tile_stats": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_f, c_f, c_vp, c_vp, c_vp, c_vp, c_f,
                                                     c_vp, c_vp, c_vp, c_vp, c_vp]),
    "oess_maxpool3x3s2_fwd_nhwc_bf16": (c_int, [c_vp, c_ll, c_int, c_int, c_int, c_int, c_vp, c_ll, c_vp, c_vp]),
    "oess_maxpool3x3s2_bwd_nhwc_bf16": (c_int, [c_vp, c_ll, c_vp, c_int, c_int, c_int, c_int, c_vp, c_ll, c_vp]),
    "oess_dropout_nhwc_bf16": (c_int, [c_vp, c_ll, c_vp, c_ll, c_ll, c_int, c_f, c_u64, c_u64, c_vp]),
    "oess_aspp_pool_fwd_f32": (c_int, [c_vp, c_f, c_vp, c_vp, c_vp, c_vp, c_vp, c_f, c_f, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "oess_aspp_pool_bwd_f32": (c_int, [c_vp, c_vp, c_f, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
}

_lib = None


class LibraryMissing(RuntimeError):
    pass


def load():
    """Load liboess.so and attach signatures.  Raises LibraryMissing (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: the HIP extension is not built. Run __graft_entry__.build() "
            "(or `make -C openess_amd/csrc`). There is no CPU fallback in the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise LibraryMissing(f"liboess.so lacks symbol {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    built = lib.oess_abi_version()
    if built != ABI_VERSION:
        raise LibraryMissing(f"{LIB_PATH} was built for C-ABI version {built}, the binding expects {ABI_VERSION}: rebuild it "
                             "(`make -C openess_amd/csrc`)")
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().oess_strerror(code).decode()
        raise RuntimeError(f"{what} failed: {msg} ({code})")
