"""ctypes binding of libigcn.so (C ABI in include/igcn.h).  No torch types cross the boundary:
only raw device pointers (tensor.data_ptr()), sizes and the hipStream_t of the current torch stream.

There is NO fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libigcn.so")

ABI_VERSION = 423        # include/igcn.h IGCN_ABI_VERSION this table was written against (tests/test_abi.py compares)

P, I, L, F, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes) ; mirrors include/igcn.h one to one
SIGNATURES = {
    "igcn_version": (I, []),
    "igcn_last_error": (ctypes.c_char_p, []),
    "igcn_configure": (I, [ctypes.c_uint, I, I]),
    "igcn_graph_plan_workspace_bytes": (Z, [L, L]),
    "igcn_graph_plan_build": (I, [L, L, P, P, P, P, P, P, P, P, P, Z, P]),
    "igcn_graph_plan_build_segmented": (I, [L, L, I, P, P, P, L, L, P, P, P, P, P, P, P, P, P]),
    "igcn_graph_plan_build_segmented_rep": (I, [L, L, I, P, P, P, L, L, P, P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P]),
    "igcn_graph_plan_tiled_workspace_bytes": (Z, [I, L, L]),
    "igcn_graph_plan_build_tiled": (I, [L, L, I, P, P, P, L, L, P, P, P, P, P, P, P, P, P, Z, P]),
    "igcn_graph_plan_replicate": (I, [L, L, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_edge_mask_fwd": (I, [L, L, I, I, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_edge_mask_bwd": (I, [L, L, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_edge_mask_reg_blocks": (I, [L, L, I, I, I]),
    "igcn_edge_mask_fwd_reg": (I, [L, L, I, I, P, P, P, P, P, P, P, P, P, P, P, P, I, F, F, F, F, F, P, P, I, P, P]),
    "igcn_edge_mask_bwd_reg": (I, [L, L, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, F, F, F, F, F, P, P, P, P, P,
                                   P, I, P, P]),
    "igcn_gcn_norm_fwd": (I, [L, L, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_gcn_norm_bwd": (I, [L, L, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_gcn_propagate_fwd": (I, [L, L, I, I, P, L, P, P, P, P, P, L, I, P]),
    "igcn_gcn_propagate_bwd_scratch_floats": (Z, [L, I]),
    "igcn_gcn_propagate_bwd": (I, [L, L, I, I, P, L, P, L, I, P, L, P, P, P, P, P, P, L, P, I, P, P, P, P]),
    "igcn_sgcn_stack_lds_bytes": (Z, [I, I, I, I, I, I]),
    "igcn_sgcn_stack_param_floats": (I, [I, I, I]),
    "igcn_sgcn_stack_fwd": (I, [L, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_sgcn_stack_bwd": (I, [L, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_sgcn_front_lds_bytes": (Z, [I, I, I, I, I]),
    "igcn_sgcn_front_fwd": (I, [L, L, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, F, F, F, F, F, P, P, P,
                                P, P, P, P]),
    "igcn_proj_bwd_supported": (I, [L, I, I]),
    "igcn_proj_bwd_blocks": (I, [L]),
    "igcn_proj_bwd_scratch_floats": (Z, [L, I]),
    "igcn_proj_bwd": (I, [L, I, I, P, P, P, P, P, P, P]),
    "igcn_proj_bwd_pair": (I, [L, I, P, P, P, P, P, P, L, I, P, P, P, P, P, P, I, P]),
    "igcn_proj_bwd_pair_bias": (I, [L, I, P, P, P, P, P, P, P, I, L, I, P, P, P, P, P, P, P, I, I, P]),
    "igcn_proj_fwd_blocks": (I, [L]),
    "igcn_proj_fwd_pair": (I, [L, I, P, P, P, P, L, I, P, P, P, P, I, P]),
    "igcn_head_bwd_supported": (I, [I, I, I]),
    "igcn_head_bwd_scratch_floats": (Z, [I, I]),
    "igcn_head_bwd_pair": (I, [I, I, I, P, P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P, P]),
    "igcn_dense_blocks_check": (I, [L, I, P, P, P]),
    "igcn_dense_sgcn_supported": (I, [I, I, I, I]),
    "igcn_dense_sgcn_ws_floats": (Z, [L, I, I, I]),
    "igcn_dense_sgcn_bwd_ws_floats": (Z, [L, I, I, I, I]),
    "igcn_dense_sgcn_reg_blocks": (I, [L, I]),
    "igcn_dense_sgcn_fwd": (I, [L, I, I, I, I, I, I, P, P, P, P, P, P, P, I, F, F, F, F, F, P, P, P, P, P, P]),
    "igcn_dense_sgcn_bwd": (I, [L, I, I, I, I, I, I, P, P, P, P, P, P, I, F, F, F, F, F, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_small_linear_bwd_scratch_floats": (Z, [L, I, I]),
    "igcn_small_linear_fwd": (I, [L, I, I, P, P, P, P, P, P]),
    "igcn_small_linear_bwd": (I, [L, I, I, P, P, P, P, P, P, P, P]),
    "igcn_small_linear_pair_fwd": (I, [L, I, I, P, P, P, P, P, I, P, P, P, P, P, P]),
    "igcn_small_linear_pair_bwd": (I, [L, I, I, P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P]),
    "igcn_snps_mask_fwd": (I, [I, I, P, P, P, P, P, P]),
    "igcn_snps_mask_bwd": (I, [I, I, P, P, P, P, P, P]),
    "igcn_head_inputs_fwd": (I, [L, I, I, I, I, P, P, P, P, P, P, P, P, P]),
    "igcn_head_inputs_bwd": (I, [L, I, I, I, I, P, P, P, P, P, P, P, P, P, P]),
    "igcn_outproj_head_inputs_fwd": (I, [L, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_head_inputs_bwd_blocks": (I, [L, I, I]),
    "igcn_head_inputs_bwd_relu": (I, [L, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, I, P, P, P]),
    "igcn_concat_cols": (I, [L, I, I, P, P, P]),
    "igcn_copy_multi": (I, [I, P, P, P, P]),
    "igcn_image_put": (I, [I, L, P, P, L, P, L, P]),
    "igcn_image_take": (I, [I, L, P, P, L, P, P]),
    "igcn_gather_batch": (I, [I, I, L, L, P, P, P, P, P, P]),
    "igcn_launch_floor": (I, [L, I, I, I, P, P]),
    "igcn_graph_pool_fwd": (I, [L, I, I, P, P, P, P]),
    "igcn_graph_pool_bwd": (I, [L, I, I, P, P, P, P]),
    "igcn_bias_grad_scratch_floats": (Z, [L, I]),
    "igcn_bias_grad": (I, [L, I, P, P, P, P, P, P]),
    "igcn_col_sums": (I, [L, I, I, P, P, P, P]),
    "igcn_bias_grad_pair": (I, [L, I, P, P, P, P, I, P, P, P, P, P, I, P, P]),
    "igcn_gemm_f32_split_k": (I, [L, L, L]),
    "igcn_gemm_f32": (I, [L, L, L, P, L, L, P, L, L, P, P, L, I, I, P, P]),
    "igcn_gemm_bf16": (I, [L, L, L, P, L, L, P, L, L, P, P, L, I, I, P, P]),
    "igcn_gemm_f32_batched_sum": (I, [L, L, L, I, P, L, L, L, P, L, L, L, P, L, P, P]),
    "igcn_gemm_f32_batched": (I, [L, L, L, I, P, L, L, L, P, L, L, L, P, L, L, I, P, P]),
    "igcn_gemm_f32_grouped": (I, [I, P, P]),
    "igcn_gemm_rider": (I, [P, I, P]),
    "igcn_gemm_rider_flush": (I, [P]),
    "igcn_rider_cancel": (I, [P]),
    "igcn_stream_pending": (I, [P]),
    "igcn_node_linear_bn_scratch_floats": (Z, [I, I, I]),
    "igcn_node_linear_bn_fwd": (I, [I, I, I, I, I, P, P, P, P, P, P, I, F, F, P, P, P, P, P, P]),
    "igcn_node_linear_bn_bwd_scratch_floats": (Z, [I, I, I, I, I]),
    "igcn_node_linear_bn_bwd": (I, [I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_node_linear_bn_pair_supported": (I, [I, I, I]),
    "igcn_node_linear_bn_pair_fwd": (I, [I, I, I, I, P, I,
                                         I, P, P, P, P, P, F, F, P, P, P, P,
                                         I, P, P, P, P, P, F, F, P, P, P, P, P, P]),
    "igcn_node_linear_bn_pair_bwd": (I, [I, I, I, I, I, P,
                                         I, P, P, P, P, P, P, P, P, P, P,
                                         I, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_bn1d_fwd": (I, [I, I, I, P, P, P, P, P, I, F, F, I, P, P, P, P, P]),
    "igcn_bn1d_bwd": (I, [I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_bn1d_fwd_supported": (I, [I, I]),
    "igcn_bn1d_fwd_slabs": (I, [I, I, I, P, I, P, P, P, P, P, I, F, F, I, P, P, P, P, P]),
    "igcn_gemm_effective_split": (I, [L, I]),
    "igcn_dropout_state_words": (I, []),
    "igcn_dropout_max_segments": (I, []),
    "igcn_dropout_masks": (I, [L, I, P, P, P, P, I, P, L, P]),
    "igcn_rider_dropout": (I, [P, L, I, P, P, P, P, I, P, L]),
    "igcn_rider_flush": (I, [P]),
    "igcn_sum_n": (I, [L, I, P, P, P]),
    "igcn_sum_n_final": (I, [L, I, P, P, P]),
    "igcn_mask_reg_blocks": (I, [L]),
    "igcn_mask_reg_fwd": (I, [L, L, L, P, P, P, F, F, F, F, F, P, P, P]),
    "igcn_mask_reg_bwd": (I, [L, L, L, P, P, P, F, F, F, F, F, P, P, P, P, P]),
    "igcn_rbf_laplacian": (I, [I, I, F, P, P, P]),
    "igcn_gram_loss_fwd": (I, [I, I, I, P, P, P, P, P]),
    "igcn_gram_loss_bwd": (I, [I, I, P, P, P, P, P]),
    "igcn_gram_loss_fwd_rbf": (I, [I, I, I, P, P, I, F, P, P, P, P]),
    "igcn_gram_loss_fwd_rbf_unit": (I, [I, I, I, P, P, I, F, P, P, P, P, P, P]),
    "igcn_attn_core_lds_bytes": (Z, [I, I, I, I, I]),
    "igcn_attn_core_fwd": (I, [I, I, I, I, I, P, P, P, P, P]),
    "igcn_attn_core_bwd_scratch_floats": (Z, [I, I, I]),
    "igcn_attn_core_bwd": (I, [I, I, I, I, I, P, P, P, P, P, P, P, P, P]),
    "igcn_attn_core_bf16_supported": (I, [I, I, I, I]),
    "igcn_attn_core_bf16_fwd": (I, [I, I, I, I, I, P, P, P, P, P]),
    "igcn_attn_core_bf16_bwd": (I, [I, I, I, I, I, P, P, P, P, P, P, P, P, P]),
    "igcn_attn_core_split_supported": (I, [I, I, I, I]),
    "igcn_attn_core_split_fwd": (I, [I, I, I, I, I, P, P, P, P, P]),
    "igcn_attn_core_split_bwd_supported": (I, [I, I, I, I]),
    "igcn_attn_core_split_bwd": (I, [I, I, I, I, I, P, P, P, P, P, P, P, P, P]),
    "igcn_spmm_fwd": (I, [I, I, I, I, L, P, P, P, P, P, P]),
    "igcn_spmm_bwd_scratch_floats": (Z, [I, I, I, I, L]),
    "igcn_spmm_bwd": (I, [I, I, I, I, L, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_spmm_bwd_dval_multi": (I, [I, P, P]),
    "igcn_spmm_fwd_strided": (I, [I, I, I, I, L, P, P, P, L, P, P, P]),
    "igcn_spmm_bwd_strided": (I, [I, I, I, I, L, P, P, P, P, P, P, P, L, P, P, P, P, P, P]),
    "igcn_go_attn_fwd": (I, [I, I, I, I, P, P, P, P, P, P, P, P, P]),
    "igcn_go_attn_bwd_scratch_floats": (Z, [I, I, I, I]),
    "igcn_go_attn_bwd_threads": (I, [I, I, I]),
    "igcn_go_attn_walk_slots": (I, [I, I, I]),
    "igcn_go_attn_walk_order": (I, [I, I, I, P, P]),
    "igcn_go_attn_bwd": (I, [I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_nodes_ln_fwd": (I, [I, I, I, I, F, P, P, P, P, P, P, P, P]),
    "igcn_nodes_ln_bwd_scratch_floats": (Z, [I, I, I]),
    "igcn_nodes_ln_bwd": (I, [I, I, I, I, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_nodes_ln_bwd_dy": (I, [I, I, I, I, P, P, P, P, P, P, P, P, P]),
    "igcn_nodes_ln_bwd_affine_multi": (I, [I, P, P]),
    "igcn_go_decode_fwd": (I, [I, I, I, I, I, P, P, P, P, P, P, P]),
    "igcn_go_attn_ln_fused_ok": (I, [I, I, I, I]),
    "igcn_go_decode_ln_fused_ok": (I, [I, I, I, I]),
    "igcn_go_ln_part_floats": (Z, [I, I]),
    "igcn_go_attn_ln_bwd": (I, [I, I, I, I, P, P, P, P, P, P, P, P, P, P, I] + [P] * 15),
    "igcn_go_decode_ln_bwd": (I, [I, I, I, I, I] + [P] * 19),
    "igcn_go_decode_bwd_scratch_floats": (Z, [I, I, I, I]),
    "igcn_go_decode_bwd": (I, [I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P]),
    "igcn_adam_step": (I, [L, P, P, P, P, P, P, F, F, F, F, P]),
    "igcn_adam_step_multi": (I, [I, P, P, P, P, F, F, F, F, P]),
    "igcn_adam_step_ticked": (I, [L, P, P, P, P, P, P, F, F, F, F, P]),
    "igcn_adam_step_multi_ticked": (I, [I, P, P, P, P, F, F, F, F, P]),
    "igcn_adam_chunk": (I, []),
    "igcn_adam_step_blocks": (I, [I, P, P, P, P, P, P, F, F, F, F, I, P]),
    "igcn_pack_grads": (I, [I, P, P, P, P, P]),
    "igcn_reduce_defer": (I, [P, I]),
    "igcn_reduce_pending": (I, []),
    "igcn_reduce_rows_final": (I, [P, L, L, I, P, P]),
    "igcn_head_loss_supported": (I, [I, I, I]),
    "igcn_head_loss_blocks": (I, [I, I]),
    "igcn_head_loss_fwd": (I, [I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, F, F, P, P, P, P, P, P, P, P, P, P]),
    "igcn_head_loss_gram_fwd": (I, [I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, F, F, P, P, P, P, P, P, P, P, P,
                                    I, I, I, P, P, I, F, P, P, P, P, P]),
    "igcn_loss_final": (I, [P, I, P, I, P, I, P, P, P]),
    "igcn_reduce_flush": (I, [P]),
    "igcn_reduce_flush_tick": (I, [P, P]),
    "igcn_comm_unique_id_bytes": (I, []),
    "igcn_comm_get_unique_id": (I, [P]),
    "igcn_comm_init": (I, [I, I, P, P]),
    "igcn_comm_allreduce": (I, [P, P, L, P]),
    "igcn_comm_destroy": (I, [P]),
    "igcn_loss_head_fwd": (I, [I, I, I, I, P, I, P, P, P, P, P, P, P, I, P, I, P, F, F, P, P, P]),
    "igcn_loss_head_bwd": (I, [I, I, I, I, P, P, P, P, P, P, P, F, F, P, P, P, P, P, P, P]),
    "igcn_loss_head_fwd_grads": (I, [I, I, I, I, P, I, P, P, P, P, P, P, P, I, P, I, P, F, F, P, P, P, P, P, P, P, P]),
    "igcn_gdc_topk_max_rois": (I, []),
    "igcn_gdc_topk": (I, [I, I, I, ctypes.c_double, P, P, P, P, P]),
    "igcn_gdc_topk_of": (I, [I, I, I, ctypes.c_double, P, L, P, P, P, P, P]),
}

_lib = None


class IgcnError(RuntimeError):
    pass


def load():
    """Load libigcn.so once.  Raises IgcnError when it has not been built (python ig-gcn_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IgcnError(f"{LIB_PATH} is missing: build it with `python ig-gcn_amd/build.py` "
                        "(there is no CPU or PyTorch fallback for the HIP path)")
    lib = ctypes.CDLL(LIB_PATH)      # torch is imported above, so libamdhip64.so.7 resolves to torch's runtime
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    got = int(lib.igcn_version())
    if got != ABI_VERSION:           # a stale libigcn.so would take shifted arguments without a word
        raise IgcnError(f"{LIB_PATH} has ABI revision {got}, this binding is written for {ABI_VERSION}: "
                        "rebuild it with `python ig-gcn_amd/build.py --force`")
    # the library's A/B switches: read from the environment HERE, once, and handed over (no getenv in a launch path)
    bits = 0
    for bit, name in enumerate(("IGCN_NO_TILED_LISTS", "IGCN_PROPAGATE_NO_LDS", "IGCN_SPMM_NO_LDS", "IGCN_GO_ATTN_CM",
                                "IGCN_DEBUG_REDUCE", "IGCN_ATTN_FP32_CORE", "IGCN_ATTN_BWD_TWICE",
                                "IGCN_ATTN_EXACT_FP32")):
        v = os.environ.get(name)
        if v is not None and (v == "1" or name in ("IGCN_NO_TILED_LISTS", "IGCN_PROPAGATE_NO_LDS", "IGCN_DEBUG_REDUCE", "IGCN_ATTN_FP32_CORE", "IGCN_ATTN_BWD_TWICE")):
            bits |= 1 << bit
    lib.igcn_configure(bits, int(os.environ.get("IGCN_GEMM_BN", "0") or 0),
                       int(os.environ.get("IGCN_ATTN_CHUNK", "0") or 0))
    _lib = lib
    return lib


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (or None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise IgcnError("libigcn operates on device memory only; got a CPU tensor")
    if not t.is_contiguous():
        raise IgcnError("libigcn needs contiguous tensors")
    return t.data_ptr()


_DEBUG_SYNC = os.environ.get("IGCN_DEBUG_SYNC", "0") == "1"


def call(name, *args):
    """Invoke an int-returning entry point; raise with igcn_last_error() on failure.

    IGCN_DEBUG_SYNC=1 (debugging a kernel fault): announce every entry point on stderr and synchronise the device
    after it, so that the last line printed names the faulting call (inside a stream capture: announced only)."""
    lib = load()
    if _DEBUG_SYNC:
        print(f"[igcn] {name} {args}", file=sys.stderr, flush=True)
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise IgcnError(f"{name} failed (rc={rc}): {lib.igcn_last_error().decode()}")
    if _DEBUG_SYNC:
        import torch
        if not torch.cuda.is_current_stream_capturing():    # (a capture refuses the synchronisation and is invalidated by it)
            torch.cuda.synchronize()


def copy_multi(pairs):
    """``dst.copy_(src)`` for every (dst, src) pair of same-shape, same-dtype contiguous DEVICE tensors as one launch
    per 16 pairs (igcn_copy_multi) on the current stream.  Anything else (a host tensor, a dtype change, a strided
    view) goes through ``Tensor.copy_``."""
    fast = []
    for dst, src in pairs:
        if (dst.is_cuda and src.is_cuda and dst.device == src.device and dst.dtype == src.dtype
                and dst.shape == src.shape and dst.is_contiguous() and src.is_contiguous()):
            if dst.numel() and dst.data_ptr() != src.data_ptr():
                fast.append((dst, src))
        else:
            dst.copy_(src, non_blocking=True)
    for k in range(0, len(fast), 16):
        chunk = fast[k:k + 16]
        n = len(chunk)
        d = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in chunk])
        s = (ctypes.c_void_p * n)(*[t.data_ptr() for _, t in chunk])
        nb = (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t, _ in chunk])
        call("igcn_copy_multi", n, d, s, nb, stream_ptr())
