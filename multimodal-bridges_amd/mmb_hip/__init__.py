"""ctypes binding of libmmb_hip.so (the C ABI declared in include/mmb_hip.h).

The library is built in-tree by mmb_hip.build (hipcc, gfx950).  There is NO fallback: if the
shared object is missing, or an op is handed a tensor that does not live on a HIP device,
the call raises -- the product path never runs on the CPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("MMB_LIB_NAME", "libmmb_hip.so"))   # developer builds: other name

c_f32p = C.c_void_p     # device pointers travel as plain addresses
c_i64p = C.c_void_p
c_i32p = C.c_void_p
c_stream = C.c_void_p

# name -> argtypes; MUST list every symbol include/mmb_hip.h declares (tests check this)
SIGNATURES = {
    "mmb_version": [],
    "mmb_mask_prepare": [C.c_int, C.c_int, C.c_void_p, C.c_int, c_f32p, c_i32p, c_i32p, c_stream],
    "mmb_source_uniform": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, C.c_uint64, C.c_uint64,
                           c_f32p, c_i64p, c_i64p, c_f32p, c_i32p, c_stream],
    "mmb_bridge_sample": [C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_float, c_f32p, c_f32p, c_f32p,
                          c_i64p, c_i64p, c_f32p, C.c_int, C.c_float, C.c_int, C.c_float,
                          c_f32p, c_f32p, c_i64p, c_stream],
    "mmb_bridge_sample_rng": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_f32p, c_i64p, C.c_int, C.c_float,
                              C.c_int, C.c_float, C.c_uint64, C.c_uint64, c_f32p, c_f32p, c_i64p, c_f32p,
                              c_i64p, c_stream],
    "mmb_telegraph_probs": [C.c_int, C.c_int, C.c_int, c_f32p, c_i64p, c_i64p, C.c_int, C.c_float, c_f32p, c_stream],
    "mmb_bridge_drift": [C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, c_f32p, c_stream],
    "mmb_euler_step": [C.c_int64, c_f32p, c_f32p, C.c_float, c_stream],
    "mmb_euler_maruyama_step": [C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_float, c_f32p, C.c_float,
                                C.c_float, c_f32p, C.c_uint64, C.c_uint64, c_stream],
    "mmb_telegraph_rate": [C.c_int, C.c_int, C.c_int, c_f32p, c_i64p, c_f32p, C.c_float, C.c_int, C.c_float,
                           c_f32p, c_stream],
    "mmb_rate_tauleap": [C.c_int, C.c_int, C.c_int, c_f32p, c_i64p, c_f32p, C.c_float, C.c_float, C.c_int,
                         C.c_float, c_f32p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, c_f32p, c_stream],
    "mmb_rate_euler_jump": [C.c_int, C.c_int, C.c_int, c_f32p, c_i64p, c_f32p, C.c_float, C.c_float, C.c_int,
                            C.c_float, c_f32p, C.c_uint64, C.c_uint64, C.c_int, c_f32p, c_stream],
    "mmb_loss_fwd_bwd": [C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                         C.c_int, c_i64p, c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream],
    "mmb_loss_fwd_bwd_noalloc": [C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                 C.c_int, c_i64p, c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream],
    "mmb_multimode_coef": [c_f32p, C.POINTER(C.c_float), c_f32p, c_stream],
    "mmb_multimode_loss": [c_f32p, c_f32p, C.POINTER(C.c_float), c_f32p, c_f32p, c_stream],
    "mmb_weight_norm": [C.c_int, c_i32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream],
    "mmb_weight_norm_bwd": [C.c_int, c_i32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream],
    "mmb_linear_cat": [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                       C.POINTER(C.c_int), c_f32p, c_f32p, C.c_int64, c_f32p, C.c_int, C.c_int, c_f32p, C.c_int64,
                       c_f32p, c_f32p, C.c_int64, c_f32p, C.c_int, C.c_float, C.c_uint64, C.c_uint64, c_f32p,
                       C.c_int64, c_f32p, C.c_int64, c_stream],
    "mmb_linear_cat_gate": [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                            C.POINTER(C.c_int), c_f32p, c_f32p, C.c_int64, C.c_int, C.c_int, c_f32p, C.c_int64, c_f32p,
                            C.c_int64, c_stream],
    "mmb_linear_cat_ln": [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                          C.POINTER(C.c_int), c_f32p, c_f32p, C.c_int64, c_f32p, C.c_int, C.c_int, c_f32p, C.c_int64,
                          c_f32p, C.c_int64, c_f32p, c_f32p, C.c_float, c_f32p, C.c_int64, c_f32p, c_stream],
    "mmb_meansum_pool": [C.c_int, C.c_int, C.c_int, c_f32p, C.c_int64, c_f32p, c_i32p, c_f32p, C.c_int64, c_stream],
    "mmb_embed": [C.c_int] * 7 + [c_f32p, C.c_float, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                  c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream],
    "mmb_act_bwd": [C.c_int64, C.c_int] + [c_f32p, C.c_int64] * 6 + [C.c_int, C.c_int, C.c_float, C.c_uint64,
                                                                      C.c_uint64, c_f32p, C.c_int64, c_stream],
    "mmb_jet_sum": [C.c_int, C.c_int, C.c_int, c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_int, c_stream],
    "mmb_pool_bwd": [C.c_int, C.c_int, C.c_int, c_f32p, C.c_int64, c_f32p, c_i32p, c_f32p, C.c_int64, C.c_int,
                     c_stream],
    "mmb_fused_epic_forward": [C.c_int, C.c_int, C.c_int, c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, c_f32p, c_i32p, c_f32p, c_i64p, c_f32p, C.c_int64, C.c_int, c_f32p, c_f32p,
                               c_stream],
    "mmb_fused_epic_train_forward": [C.c_int, C.c_int, C.c_int, c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, c_f32p, c_i32p, c_f32p, c_i64p, c_f32p, C.c_int64, C.c_void_p,
                                     C.c_int, C.c_float, C.c_uint64, C.c_uint64, c_f32p, c_f32p, c_i32p, c_stream],
    "mmb_fused_epic_train_backward": [C.c_int, C.c_int, C.c_int, c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p,
                                      c_i32p, C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint64, c_i32p, c_stream],
    "mmb_grad_weight_smallk": [C.c_int, c_f32p, C.c_int64, C.c_int, c_f32p, C.c_int64, C.c_int, c_f32p, C.c_int64, c_f32p,
                               c_stream],
    "mmb_weight_norm_multi": [C.c_int, C.c_void_p, c_stream],
    "mmb_weight_norm_bwd_multi": [C.c_int, C.c_void_p, c_stream],
    "mmb_pack_params": [c_f32p, c_i64p, c_i64p, C.c_int64, c_i64p, C.c_int64, c_f32p, c_stream],
    "mmb_grad_weight_rows_multi": [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, c_stream],
    "mmb_fused_uni_train_forward": [C.c_int, C.c_int, C.c_int, c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, c_f32p, c_i32p,
                                    c_i64p, c_f32p, C.c_int64, C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint64,
                                    c_f32p, c_i32p, c_stream],
    "mmb_fused_uni_train_backward": [C.c_int, C.c_int, C.c_int, c_f32p, c_i32p, C.c_int, C.c_int, c_f32p, c_i32p,
                                     C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint64, c_i32p, c_stream],
    "mmb_fused_epic_generate": [C.c_int, C.c_int, C.c_int, c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, c_f32p, c_i32p, c_f32p, c_i64p, C.c_int, c_f32p, c_f32p, c_f32p,
                                C.c_int, C.c_float,
                                C.c_int, C.c_float, c_f32p, C.c_uint64, C.c_uint64, C.c_int, C.c_float, c_f32p,
                                c_i32p, c_i32p, C.c_int, c_stream],
    "mmb_jet_schedule": [C.c_int, c_i32p, c_i32p, c_stream],
    "mmb_mask_rows": [C.c_int, C.c_int, c_f32p, c_i32p, c_i32p, c_stream],
    "mmb_jet_observables": [C.c_int, C.c_int, C.c_int, c_f32p, c_i64p, c_f32p, c_f32p, c_stream],
    "mmb_postprocess": [C.c_int64, C.c_int, C.c_int, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p, C.c_int, c_stream],
    "mmb_postprocess_onehot": [C.c_int64, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p,
                               C.c_int, c_stream],
    "mmb_grad_weight_rows": [C.c_int, c_f32p, C.c_int64, C.c_int, c_f32p, C.c_int64, C.c_int, c_f32p, C.c_int64, c_f32p,
                             C.c_int, C.c_int64, C.c_int64, c_stream],
    "mmb_weight_norm_bwd_rep": [C.c_int, c_i32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int64, c_f32p,
                                C.c_int64, c_f32p, c_f32p, c_f32p, c_stream],
    "mmb_fused_uni_forward": [C.c_int, C.c_int, C.c_int, c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, c_f32p, c_i32p,
                              c_i64p, c_f32p, C.c_int64, c_f32p, c_stream],
    "mmb_fused_uni_generate": [C.c_int, C.c_int, C.c_int, c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, c_f32p, c_i32p,
                               c_i64p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_float, C.c_int, C.c_float,
                               c_f32p, C.c_uint64, C.c_uint64, c_i32p, c_i32p, C.c_int, c_stream],
    "mmb_attention_fwd": [C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int64, c_f32p, C.c_int, c_f32p, C.c_int64,
                          c_f32p, c_stream],
    "mmb_attention_bwd": [C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int64, c_f32p, C.c_int, c_f32p, C.c_int64,
                          c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, c_stream],
    "mmb_attention_dropout_fwd": [C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int64, c_f32p, C.c_int, c_f32p,
                                  C.c_int64, c_f32p, C.c_float, C.c_uint64, C.c_uint64, c_i32p, c_stream],
    "mmb_attention_dropout_bwd": [C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int64, c_f32p, C.c_int, c_f32p,
                                  C.c_int64, c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_float, C.c_uint64,
                                  C.c_uint64, c_i32p, c_stream],
    "mmb_attention_dropout_mask": [C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, C.c_float, C.c_uint64, C.c_uint64, c_f32p,
                                   c_stream],
    "mmb_transformer_tail": [C.c_int64, C.c_int, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p,
                             C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_float, C.c_float,
                             c_i32p, c_stream],
    "mmb_layernorm_fwd": [C.c_int64, C.c_int, c_f32p, C.c_int64, c_f32p, c_f32p, C.c_float, c_f32p, C.c_int64, c_f32p,
                          c_stream],
    "mmb_layernorm_bwd": [C.c_int64, C.c_int, c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, C.c_int64, c_f32p, C.c_int64,
                          C.c_int, c_f32p, c_f32p, c_stream],
    "mmb_sum_replicas": [C.c_int64, C.c_int, C.c_int64, c_f32p, c_f32p, C.c_int, c_stream],
    "mmb_adamw_clip_step": [C.c_int64, C.c_int64, c_f32p, c_f32p, c_f32p, c_f32p, C.c_float, C.c_float, C.c_float,
                            C.c_float, C.c_float, C.c_int64, C.c_float, c_f32p, c_f32p, C.c_int, c_stream],
    "mmb_toy_rows": [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int64, c_f32p, c_i64p, c_f32p, c_f32p,
                     c_stream],
    "mmb_toy_embedding_bwd": [C.c_int64, C.c_int, C.c_int, c_i64p, c_f32p, C.c_int64, C.c_int, c_f32p, c_stream],
    "mmb_toy_sample": [C.c_int, C.c_int64, C.c_int, C.c_float, C.c_uint64, C.c_uint64, c_f32p, c_i64p, c_stream],
    "mmb_aoj_features": [C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_i64p, c_i64p, c_stream],
    "mmb_toy_generate": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.POINTER(C.c_void_p), c_f32p, c_f32p,
                         c_i64p, C.c_int, c_f32p, C.c_float, C.c_int, C.c_float, c_f32p, C.c_uint64, C.c_uint64, C.c_int,
                         C.c_float, c_f32p, c_stream],
    "mmb_grad_weight": [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                        C.POINTER(C.c_int), c_f32p, c_f32p, C.c_int64, C.c_int, c_f32p, C.c_int64, c_f32p, c_stream],
}

_lib = None


class MMBError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMBError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        h.mmb_last_error.restype = C.c_char_p
        h.mmb_last_error.argtypes = []
        for name, args in SIGNATURES.items():
            if "MMB_LIB_NAME" in os.environ and not hasattr(h, name):
                continue                    # developer A/B build that predates an entry point (never the shipped library)
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = C.c_int
        _lib = h
    return _lib


# Optional in-process kernel timing for bench.py: set PROFILE to a list and every C-ABI call
# is bracketed by HIP events recorded on the stream it is enqueued on (torch's current one).
PROFILE = None
PROFILE_META = None


def call(name, *args):
    h = lib()
    if PROFILE is not None:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(h, name)(*args)
        e1.record()
        PROFILE.append((name, PROFILE_META, e0, e1))
    else:
        rc = getattr(h, name)(*args)
    if rc != 0:
        raise MMBError(f"{name} failed (rc={rc}): {h.mmb_last_error().decode()}")
    return rc
