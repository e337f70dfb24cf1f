"""ctypes binding of libglnn_hip.so (the C ABI declared in include/glnn_hip.h).

There is NO fallback: if the shared library is missing or does not export a symbol the import of
this module's `lib()` raises, and every op in glnn_amd.ops raises on non-CUDA tensors.  The library is
built in-tree by `python __graft_entry__.py` (hipcc --offload-arch=gfx950)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GLNN_LIB_PATH") or os.path.join(_HERE, "libglnn_hip.so")   # override: A/B of kernel variants

c_i64, c_int, c_f32, c_vp, c_u32 = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint32

# name -> argtypes, exactly the prototypes of include/glnn_hip.h (pointers as void*)
SIGNATURES = {
    "glnn_abi_version": [],
    "glnn_device_info": [c_vp, c_vp, c_vp, c_int],
    "glnn_struct_bytes": [c_int],
    "glnn_spmm_csr_f32": [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp,
                          c_int, c_vp, c_i64, c_vp],
    "glnn_packed_weight_floats": [c_int, c_int],
    "glnn_pack_weight_f32": [c_vp, c_i64, c_int, c_int, c_vp, c_vp],
    "glnn_sage_fused_f32": [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_int, c_vp, c_vp, c_int, c_vp,
                            c_i64, c_vp, c_int, c_vp, c_i64, c_vp, c_vp],
    "glnn_hub_row_threshold": [],
    "glnn_hub_segment_edges": [],
    "glnn_spmm_csr_plan_f32": [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp,
                               c_int, c_vp, c_i64, c_vp, c_vp],
    "glnn_sage_fused_plan_f32": [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_int, c_vp, c_vp, c_int, c_vp,
                                 c_i64, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_vp],
    "glnn_sage_fused_chunks_f32": [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_int, c_vp, c_vp, c_int, c_vp,
                                   c_i64, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp],
    "glnn_spmm_csr_chunks_f32": [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp,
                                 c_int, c_vp, c_i64, c_vp, c_vp, c_vp],
    "glnn_signal_alloc": [c_vp],
    "glnn_signal_free": [c_vp],
    "glnn_signal_read": [c_vp, c_vp],
    "glnn_stream_wait_value32": [c_vp, c_vp, c_u32],
    "glnn_degrees_f32": [c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp, c_vp, c_vp],
    "glnn_gemm_f32": [c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_u32, c_i64, c_int, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int,
                      c_vp, c_i64, c_vp, c_i64, c_vp],
    "glnn_gemm_tn_f32": [c_vp, c_i64, c_i64, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_u32, c_int, c_vp, c_i64, c_vp, c_vp,
                         c_i64, c_vp],
    "glnn_softmax_loss_f32": [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_f32, c_vp, c_i64,
                              c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp],
    "glnn_log_softmax_f32": [c_vp, c_i64, c_i64, c_int, c_vp, c_i64, c_vp],
    "glnn_classifier_loss_f32": [c_vp, c_i64, c_vp, c_vp, c_f32, c_u32, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_int,
                                 c_vp, c_vp, c_vp, c_i64, c_vp, c_f32, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp],
    "glnn_linear_bn_stats_f32": [c_vp, c_i64, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp,
                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp],
    "glnn_bn_stats_f32": [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                          c_vp, c_vp, c_i64, c_vp],
    "glnn_bn_relu_bwd_f32": [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_u32, c_vp, c_i64,
                             c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "glnn_col_sum_f32": [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_i64, c_vp],
    "glnn_adam_step_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i64, c_vp],
    "glnn_mlp_fwd_bwd_f32": [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_vp, c_f32, c_vp, c_vp],
    "glnn_mlp_train_step_f32": [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_vp, c_f32, c_vp, c_vp, c_vp],
    "glnn_sage_fwd_bwd_f32": [c_vp, c_vp],
    "glnn_sage_train_step_f32": [c_vp, c_vp, c_vp],
    "glnn_sage_step_ws_bn_floats": [c_i64, c_int],
    "glnn_act_fwd_f32": [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_f32, c_u32, c_vp, c_i64, c_vp],
    "glnn_norm_drop_fwd_f32": [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_int, c_f32, c_u32, c_vp, c_i64, c_vp],
    "glnn_bn_bwd_f32": [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_u32, c_vp, c_i64,
                        c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "glnn_layernorm_fwd_f32": [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_f32, c_int, c_f32, c_u32, c_vp, c_i64, c_vp, c_vp, c_vp],
    "glnn_layernorm_bwd_workspace_floats": [c_i64, c_int],
    "glnn_layernorm_bwd_f32": [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_u32, c_vp, c_i64,
                               c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "glnn_dropout_mask_u8": [c_i64, c_int, c_f32, c_u32, c_vp, c_vp],
    "glnn_sample_neighbors": [c_vp, c_vp, c_vp, c_i64, c_int, c_u32, c_vp, c_vp, c_vp],
    "glnn_block_workspace_bytes": [c_i64, c_i64],
    "glnn_block_build": [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "glnn_block_build_ids": [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp],
    "glnn_csr_transpose_workspace_bytes": [c_i64, c_i64],
    "glnn_csr_transpose": [c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_i64, c_vp],
    "glnn_gather_rows_f32": [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_vp],
    "glnn_scatter_rows_f32": [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_vp],
}

MLP_MAX_LAYERS = 8
MLP_COUNTERS = 1024
_F = ctypes.c_void_p * MLP_MAX_LAYERS


ABI_VERSION = 12
# glnn_exchange_fn: int (*)(void* ctx, const float* send, float* recv, int64_t floats, void* stream)
GRAD_READY_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)


class AdamDesc(ctypes.Structure):
    """glnn_adam_desc of include/glnn_hip.h (field for field)."""
    _fields_ = [("params", ctypes.c_void_p), ("grads", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("sizes", ctypes.c_void_p), ("grads_host", ctypes.c_void_p), ("num_tensors", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("max_size", c_i64), ("lr", c_f32), ("beta1", c_f32), ("beta2", c_f32), ("eps", c_f32), ("weight_decay", c_f32),
                ("reserved2", ctypes.c_int32), ("step", c_i64)]


class HubPlanDesc(ctypes.Structure):
    """glnn_hub_plan of include/glnn_hip.h (field for field)."""
    _fields_ = [("rows", c_vp), ("seg_ptr", c_vp), ("n_hub", ctypes.c_int32), ("n_seg", ctypes.c_int32), ("slab", c_vp),
                ("ld_slab", c_i64), ("slab_rows", c_i64)]


MAX_CHUNKS = 8


class ChunkSignalsDesc(ctypes.Structure):
    """glnn_chunk_signals of include/glnn_hip.h (field for field)."""
    _fields_ = [("n_chunks", ctypes.c_int32), ("row_start", c_i64 * (MAX_CHUNKS + 1)), ("self_row", c_i64 * MAX_CHUNKS),
                ("out_row", c_i64 * MAX_CHUNKS), ("arrivals", c_vp), ("signal", c_vp * MAX_CHUNKS), ("epoch", ctypes.c_uint32)]


class MlpStepDesc(ctypes.Structure):
    """glnn_mlp_step_desc of include/glnn_hip.h (field for field)."""
    _fields_ = ([("num_layers", ctypes.c_int32), ("batchnorm", ctypes.c_int32), ("dims", ctypes.c_int32 * (MLP_MAX_LAYERS + 1)),
                 ("dropout_p", c_f32), ("bn_eps", c_f32), ("bn_momentum", c_f32), ("max_batch", c_i64)] +
                [(n, _F) for n in ("w", "b", "gw", "gb", "gamma", "beta", "ggamma", "gbeta", "running_mean", "running_var", "nbt",
                                   "mean", "rstd", "a_scale", "a_shift", "z")] +
                [("ldz", c_i64 * MLP_MAX_LAYERS),
                 ("logits", c_vp), ("ld_logits", c_i64), ("dlogits", c_vp), ("ld_dlogits", c_i64),
                 ("da", c_vp), ("ld_da", c_i64), ("dz", c_vp), ("ld_dz", c_i64),
                 ("ws_bn", c_vp), ("ws_bn_floats", c_i64), ("ws_tn", c_vp), ("ws_tn_floats", c_i64),
                 ("ws_gemm", c_vp), ("ws_gemm_floats", c_i64), ("ws_loss", c_vp), ("ws_loss_floats", c_i64),
                 ("loss_out", c_vp), ("loss_accum", c_vp),
                 ("world", ctypes.c_int32), ("rank", ctypes.c_int32), ("exchange", EXCHANGE_FN), ("exchange_ctx", c_vp),
                 ("sync_send", c_vp), ("sync_recv", c_vp), ("sync_rows", c_vp), ("sync_counters", c_vp),
                 ("act", _F), ("ld_act", c_i64 * MLP_MAX_LAYERS), ("xb", c_vp), ("ld_xb", c_i64),
                 ("grad_ready", GRAD_READY_FN), ("grad_ready_ctx", c_vp),
                 ("dz2", c_vp), ("ld_dz2", c_i64)])


SAGE_MAX_LAYERS = 8


class SageLayer(ctypes.Structure):
    """glnn_sage_layer of include/glnn_hip.h (field for field)."""
    _fields_ = ([("indptr", c_vp), ("indices", c_vp), ("n_dst", c_i64), ("n_src", c_i64), ("nnz", c_i64), ("self_rows", c_vp)] +
                [(n, c_vp) for n in ("w", "b", "gw", "gb", "gamma", "beta", "ggamma", "gbeta", "running_mean", "running_var", "nbt",
                                     "mean", "rstd", "a_scale", "a_shift")] +
                [("agg", c_vp), ("ld_agg", c_i64), ("z", c_vp), ("ldz", c_i64), ("h", c_vp), ("ldh", c_i64),
                 ("t_indptr", c_vp), ("t_indices", c_vp), ("inv_deg", c_vp), ("tr_ws", c_vp), ("tr_ws_bytes", c_i64), ("drop_seed", c_u32)])


class SageStepDesc(ctypes.Structure):
    """glnn_sage_step_desc of include/glnn_hip.h (field for field)."""
    _fields_ = [("num_layers", ctypes.c_int32), ("batchnorm", ctypes.c_int32), ("dims", ctypes.c_int32 * (SAGE_MAX_LAYERS + 1)),
                ("dropout_p", c_f32), ("bn_eps", c_f32), ("bn_momentum", c_f32), ("lamb", c_f32),
                ("layer", SageLayer * SAGE_MAX_LAYERS),
                ("x", c_vp), ("ldx", c_i64), ("x_rows", c_i64), ("labels", c_vp), ("label_rows", c_vp),
                ("dlogits", c_vp), ("ld_dlogits", c_i64), ("dagg", c_vp), ("ld_dagg", c_i64), ("dh", c_vp), ("ld_dh", c_i64),
                ("ws_bn", c_vp), ("ws_bn_floats", c_i64), ("ws_tn", c_vp), ("ws_tn_floats", c_i64), ("ws_gemm", c_vp), ("ws_gemm_floats", c_i64),
                ("ws_loss", c_vp), ("ws_loss_floats", c_i64), ("loss_out", c_vp), ("loss_accum", c_vp)]


_lib = None


class GlnnError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GlnnError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # torch first: its wheel bundles its own HIP runtime, and a process must end up with ONE of them.  Loading this
        # library before torch binds it to /opt/rocm's libamdhip64 while torch later initialises its bundled copy -- every
        # launch from here then fails with "no ROCm-capable device is detected" (seen as build(); smoke() in one process).
        import torch  # noqa: F401
        h = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError => missing export: fail loudly
            fn.argtypes = argtypes
            fn.restype = c_int
        h.glnn_packed_weight_floats.restype = c_i64
        h.glnn_struct_bytes.restype = c_i64
        h.glnn_block_workspace_bytes.restype = c_i64
        h.glnn_csr_transpose_workspace_bytes.restype = c_i64
        h.glnn_layernorm_bwd_workspace_floats.restype = c_i64
        h.glnn_sage_step_ws_bn_floats.restype = c_i64
        h.glnn_last_error.argtypes = []
        h.glnn_last_error.restype = ctypes.c_char_p
        h.glnn_reload_options.argtypes = []
        h.glnn_reload_options.restype = None
        if h.glnn_abi_version() != ABI_VERSION:
            raise GlnnError(f"{LIB_PATH}: ABI version {h.glnn_abi_version()} != {ABI_VERSION} expected by this package; rebuild")
        for which, mirror in ((0, MlpStepDesc), (1, SageStepDesc), (2, SageLayer), (3, AdamDesc), (4, HubPlanDesc), (5, ChunkSignalsDesc)):
            if h.glnn_struct_bytes(which) != ctypes.sizeof(mirror):
                raise GlnnError(f"{LIB_PATH}: sizeof({mirror.__name__}) is {h.glnn_struct_bytes(which)} in the library, "
                                f"{ctypes.sizeof(mirror)} in this binding")
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().glnn_last_error().decode(errors="replace")
        raise GlnnError(f"{what} failed (status {rc}): {msg}")
