"""ctypes binding of libmatcha_hip.so (the C ABI declared in include/matcha_hip.h).

There is no CPU fallback: if the library is missing, every entry point raises.  PyTorch is used by the
callers only for device memory (``tensor.data_ptr()``) and streams.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MATCHA_HIP_LIB") or os.path.join(_HERE, "lib", "libmatcha_hip.so")   # override: A/B builds

MAX_L = 8
N_HEAD = 8

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_BIAS, EPI_TANH, EPI_DROPOUT, EPI_ROWMASK, EPI_RESIDUAL, EPI_DTANH, EPI_ACCUM = 1, 2, 4, 8, 16, 32, 64

PROF = dict(gemm_nt=1, gemm_nn=2, gemm_tn=3, attn_fwd=4, attn_bwd=5, embed_fwd=6, embed_scatter=7, ln3_fwd=8, ln3_bwd=9,
            head_fwd=10, head_bwd=11, adamw=12, neg_sample=13, adj_encode=14, gather_rows=15, fused_fwd=16, fused_bwd=17, front_fwd=18, front_bwd=19,
            adj_recon=20, adj_bwd=21)

_fp = C.c_void_p  # device pointers travel as void*


class Shape(C.Structure):
    _fields_ = [("d", C.c_int32), ("n_attr", C.c_int32), ("n_nodes", C.c_int32), ("n_chrom", C.c_int32),
                ("mode", C.c_int32), ("max_bins", C.c_int32)]


TENSOR_FIELDS = ["table", "adj_w0", "adj_w1", "recon_w", "recon_b", "attr_w", "attr_b", "next_w", "next_b",
                 "ln_q_g", "ln_q_b", "ln_k_g", "ln_k_b", "ln_v_g", "ln_v_b", "w_q", "w_k", "w_v", "fc1_w", "fc1_b",
                 "pff0_w", "pff0_b", "pff1_w", "pff1_b", "pff_ln_g", "pff_ln_b", "ln1_g", "ln1_b", "ln2_g", "ln2_b",
                 "cls_w", "cls_b"]


class Tensors(C.Structure):
    _fields_ = [(n, _fp) for n in TENSOR_FIELDS]


class Frozen(C.Structure):
    _fields_ = [("attr_table", _fp), ("bounds", _fp), ("feats", _fp), ("feat_off", _fp), ("inter", _fp),
                ("bounds_host", _fp), ("feat_row_pad", C.c_int32), ("attr_mode", C.c_int32), ("attr_ld", C.c_int32),
                ("attr_scale", C.c_float), ("attr_bounds", _fp)]


class StepOpts(C.Structure):
    _fields_ = [("training", C.c_int32), ("random_chrom", C.c_int32), ("p_drop_adj", C.c_float),
                ("p_drop_fc1", C.c_float), ("p_drop_pff", C.c_float), ("alpha", C.c_float), ("beta", C.c_float),
                ("seed", _fp), ("forward_only", C.c_int32), ("loss_in_forward", C.c_int32), ("status", _fp),
                ("sparse_table_grad", C.c_int32), ("deterministic", C.c_int32), ("encoder_done_event", _fp),
                ("random_chrom_dev", _fp)]


class RaggedView(C.Structure):
    _fields_ = [("row_off", _fp), ("tok_slot", _fp), ("tok_id", _fp), ("tok_key", _fp), ("tok_pos", _fp), ("count", _fp),
                ("tile_meta", _fp), ("tiles_cap", C.c_int64), ("half_meta", _fp), ("halves_cap", C.c_int64), ("tok_tile", _fp)]


class GemmEpilogue(C.Structure):
    _fields_ = [("flags", C.c_int32), ("bias", _fp), ("residual", _fp), ("aux", _fp), ("row_ids", _fp), ("seed", _fp),
                ("stream_id", C.c_int32), ("p_drop", C.c_float), ("aux_scale", C.c_float)]


# name -> (restype, argtypes)   -- every symbol include/matcha_hip.h declares
_I64, _I32, _SZ, _D = C.c_int64, C.c_int32, C.c_size_t, C.c_double
SIGNATURES = {
    "matcha_abi_version": (C.c_int, []),
    "matcha_last_error": (C.c_char_p, []),
    "matcha_device_count": (C.c_int, []),
    "matcha_profile_select": (C.c_int, [_I32]),
    "matcha_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "matcha_launch_log": (C.c_int, [_I32]),
    "matcha_launch_log_read": (C.c_int, [C.c_char_p, _SZ]),
    "matcha_workspace_bytes": (_SZ, [C.POINTER(Shape), _I64, _I32]),
    "matcha_workspace_bytes_forward": (_SZ, [C.POINTER(Shape), _I64, _I32]),
    "matcha_forward": (C.c_int, [C.POINTER(Shape), C.POINTER(Tensors), C.POINTER(Frozen), C.POINTER(StepOpts), _fp, _I64,
                                 _I32, _fp, _fp, _fp, _fp, _fp, _SZ, _fp]),
    "matcha_backward": (C.c_int, [C.POINTER(Shape), C.POINTER(Tensors), C.POINTER(Frozen), C.POINTER(StepOpts), _fp, _I64,
                                  _I32, _fp, _fp, _fp, _fp, C.POINTER(Tensors), _fp, _fp, _SZ, _fp]),
    "matcha_node_embeddings": (C.c_int, [C.POINTER(Shape), C.POINTER(Tensors), C.POINTER(Frozen), _fp, _I64, _fp, _fp, _SZ, _fp, _fp]),
    "matcha_get_embedding": (C.c_int, [C.POINTER(Shape), C.POINTER(Tensors), C.POINTER(Frozen), C.POINTER(StepOpts), _fp, _I64, _I32,
                                       _fp, _fp, _fp, _fp, _fp, _SZ, _fp]),
    "matcha_table_grad_rows": (C.c_int, [C.POINTER(Shape), _I64, _I32, _fp, _SZ, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "matcha_scatter_rows_workspace_bytes": (_SZ, [_I64, _I32, _I32]),
    "matcha_scatter_rows": (C.c_int, [_fp, _fp, _I64, _I32, _I32, _fp, _fp, _SZ, _fp]),
    "matcha_ragged_plan_bytes": (_SZ, [_I64, _I32]),
    "matcha_ragged_plan": (C.c_int, [_fp, _I64, _I32, _I32, _fp, _fp, _SZ, C.POINTER(RaggedView), _fp]),
    "matcha_random_chrom_dev_supported": (C.c_int, [C.POINTER(Shape), C.POINTER(Frozen)]),
    "matcha_set_option": (C.c_int, [C.c_char_p, _I32]),
    "matcha_get_option": (C.c_int32, [C.c_char_p]),
    "matcha_adamw_step": (C.c_int, [_fp, _fp, _fp, _fp, _I64, _fp, _I32, _fp, _fp, _fp, _fp, _D, _D, _D, _D, _D, _D, _fp]),
    "matcha_hashset_bytes": (_SZ, [_I64]),
    "matcha_hashset_build": (C.c_int, [_fp, _SZ, _fp, _I64, _I32, _fp]),
    "matcha_hashset_contains": (C.c_int, [_fp, _fp, _I32, _fp, _I64, _I32, _fp, _fp]),
    "matcha_step_select": (C.c_int, [_fp, _fp, _I64, _I32, _fp, _I32, _fp, _fp, _fp, _I64, _fp, _fp, _fp, _fp]),
    "matcha_step_record": (C.c_int, [_fp, _fp, _fp, _I64, _I32, _fp, _I64, _fp, _fp, _fp, _fp]),
    "matcha_neg_sample": (C.c_int, [_fp, _fp, _I64, _I32, _fp, _I64, _I32, _I32, _I32, _fp, _I32, _fp, _I32, _fp, _fp, _fp, _fp]),
    "matcha_gemm_tn_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "matcha_gemm": (C.c_int, [_I32, _fp, _fp, _fp, _I64, _I64, _I64, C.POINTER(GemmEpilogue), _fp, _fp, _fp, _SZ, _fp]),
    "matcha_embed_fwd": (C.c_int, [_fp, _I64, _I32, _fp, _fp, _fp, _I32, _fp, _fp, _fp, _fp]),
    "matcha_embed_scatter_bwd": (C.c_int, [_fp, _I64, _I32, _fp, _fp, _fp]),
    "matcha_ln3_fwd": (C.c_int, [_fp, _I64, _I32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "matcha_attn_bwd_workspace_bytes": (_SZ, [_I64, _I32]),
    "matcha_attn_fwd": (C.c_int, [_fp, _fp, _fp, _fp, _I64, _I32, _I32, _fp, _fp, _fp]),
    "matcha_attn_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _I64, _I32, _I32, _fp, _fp, _fp, _fp, _SZ, _fp]),
    "matcha_kmer_workspace_bytes": (_SZ, [_I64, _I32, _I32]),
    "matcha_kmer_generate": (C.c_int, [_fp, _fp, _fp, _I64, _I64, _I32, _I32, _I32, _I32, _fp, _fp, _I64, _fp, _fp, _SZ, _fp]),
    "matcha_quantile_workspace_bytes": (_SZ, [_I64]),
    "matcha_quantile_uniform": (C.c_int, [_fp, _I64, _I32, _fp, _fp, _fp, _SZ, _fp]),
    "matcha_pixels_to_adj": (C.c_int, [_fp, _fp, _fp, _I64, _fp, _I64, _fp, _I32, _fp, _fp, _fp]),
    "matcha_corrcoef_workspace_bytes": (_SZ, [_I32]),
    "matcha_corrcoef_block": (C.c_int, [_fp, _I64, _I32, _fp, _fp, _SZ, _fp]),
    "matcha_zscore_rows": (C.c_int, [_fp, _I64, _I64, _fp]),
}

FEAT_ROW_PAD = 64           # adj front end: feature rows padded to this many floats (matcha_frozen.feat_row_pad)
ABI_VERSION = 7             # MATCHA_ABI_VERSION of include/matcha_hip.h

_lib = None


class MatchaHipError(RuntimeError):
    pass


def load():
    """dlopen libmatcha_hip.so once; raise (loudly) when it is absent -- there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MatchaHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  matcha_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.matcha_abi_version() != ABI_VERSION:
        raise MatchaHipError("libmatcha_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().matcha_last_error().decode("utf-8", "replace")
        raise MatchaHipError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


STATUS_BAD_ID, STATUS_BAD_CHROM = 1, 2      # bits of status[0] (include/matcha_hip.h)


def set_option(name: str, value: int = 1) -> int:
    """Flip one of the library's process-wide A/B switches (matcha_set_option); returns the previous value."""
    lib = load()
    old = lib.matcha_get_option(name.encode())
    check(lib.matcha_set_option(name.encode(), int(value)), "matcha_set_option")
    return old


class option:
    """``with _lib.option("disable_fused"): ...`` -- switch on for the block, restore afterwards (tests)."""

    def __init__(self, name: str, value: int = 1):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = set_option(self.name, self.value)
        return self

    def __exit__(self, *a):
        set_option(self.name, self.old)


class launch_log:
    """``with _lib.launch_log() as log: ...; log.counts`` -- {kernel name: launches} of every kernel the library launched inside
    the block (matcha_launch_log).  The parity tests assert the kernel set with it."""

    def __enter__(self):
        self.counts = {}
        check(load().matcha_launch_log(1), "matcha_launch_log")
        return self

    def __exit__(self, *a):
        lib = load()
        lib.matcha_launch_log(0)
        buf = C.create_string_buffer(1 << 14)
        check(lib.matcha_launch_log_read(buf, len(buf)), "matcha_launch_log_read")
        for line in buf.value.decode().splitlines():
            name, n = line.rsplit(" ", 1)
            self.counts[name] = int(n)


def raise_on_status(status_host, what: str):
    """Translate a status word read back from the device (list of 4 ints) into the exception the reference raises."""
    if status_host[0] & STATUS_BAD_ID:
        raise IndexError(f"{what}: node id outside [0, n_nodes] (the reference's nn.Embedding raises 'index out of range in self')")
    if status_host[0] & STATUS_BAD_CHROM:
        raise KeyError(f"{what}: a node without a chromosome (node2chrom < 0) reached the negative sampler")
