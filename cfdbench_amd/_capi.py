"""ctypes binding of include/cfdbench_amd.h.

``CApi`` wraps an already-loaded shared library; every method takes raw device addresses (ints) and
raises ``CfdError`` on a non-zero status.  The product loads ``cfdbench_amd/_C/libcfdbench_amd.so``
through ``cfdbench_amd._lib`` (and fails loudly when it is missing); the CPU tests load the same
sources built against the SIMT emulator (tests/emul) through this very class.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

CFD_MAX_LAYERS = 16


class CfdError(RuntimeError):
    pass


class FnoShape(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "B", "H", "W", "in_chan", "out_chan", "n_case_params", "hidden", "num_layers", "modes1", "modes2", "head")]


class FnoParams(C.Structure):
    _fields_ = [
        ("fc0_w", C.c_void_p), ("fc0_b", C.c_void_p),
        ("spec_w1", C.c_void_p * CFD_MAX_LAYERS), ("spec_w2", C.c_void_p * CFD_MAX_LAYERS),
        ("w0_w", C.c_void_p * CFD_MAX_LAYERS), ("w0_b", C.c_void_p * CFD_MAX_LAYERS),
        ("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p), ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p),
    ]


class FfnStackArgs(C.Structure):
    """``cfd_ffn_stack_args`` (include/cfdbench_amd.h): one Linear(+activation) stack of a multi-stack launch; w / b / y / z / gw / gb point
    at host arrays of ``L`` device pointers, ``dims`` at ``L + 1`` ints."""
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("b", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("R", C.c_int),
                ("dims", C.c_void_p), ("L", C.c_int), ("act", C.c_int), ("act_last", C.c_int),
                ("gy", C.c_void_p), ("gw", C.c_void_p), ("gb", C.c_void_p), ("gx", C.c_void_p), ("ws", C.c_void_p)]


_P = C.c_void_p
ABI_VERSION = 601  # include/cfdbench_amd.h: CFD_ABI_VERSION
_I = C.c_int
_F = C.c_float
_Z = C.c_size_t

_SIGS = {
    "cfd_version": (C.c_int, []),
    "cfd_last_error": (C.c_char_p, []),
    "cfd_tune_set": (_I, [C.c_char_p, _I]),
    "cfd_prof_begin": (_I, []),
    "cfd_prof_end": (_I, [C.c_char_p, _Z]),
    "cfd_plan_create": (_I, [_I, _I, _I, _I, C.POINTER(_P)]),
    "cfd_plan_destroy": (None, [_P]),
    "cfd_spectral_dft": (_I, [_P, _P, _P, _I, _I, _P]),
    "cfd_spectral_mix": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfd_spectral_idft": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "cfd_spectral_wgrad_workspace_bytes": (_Z, [_P, _I, _I, _I]),
    "cfd_spectral_wgrad": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfd_spectral_mix_adj_wgrad": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfd_spectral_conv2d_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfd_spectral_conv2d_bwd_workspace_bytes": (_Z, [_P, _I, _I, _I]),
    "cfd_spectral_conv2d_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfd_fno_block_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfd_fno_block_bwd_input": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfd_chanmix": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cfd_chan_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "cfd_chan_wgrad": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfd_fno_stem_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfd_fno_stem_bwd_workspace_bytes": (_Z, [_P, _I, _I, _I, _I]),
    "cfd_fno_stem_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfd_fno_head_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "cfd_fno_head_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cfd_fno_head_bwd": (_I, [_P] * 15 + [_I, _I, _I, _I, _I, _I, _P]),
    "cfd_fno_head_train": (_I, [_P] * 16 + [_I, _I, _I, _I, _I, _I, _P]),
    "cfd_loss_workspace_bytes": (_Z, [_Z]),
    "cfd_masked_loss_sums": (_I, [_P, _P, _P, _P, _Z, _P]),
    "cfd_loss_sums_bwd": (_I, [_P, _P, _P, _P, _P, _Z, _P]),
    "cfd_loss_scores": (_I, [_P, _P, _P]),
    "cfd_loss_scores_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "cfd_mse_loss_fwd": (_I, [_P, _P, _P, _P, _P, _Z, _P]),
    "cfd_mse_loss_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "cfd_mse_loss_fwd_ld": (_I, [_P, _P, _P, _P, _P, _Z, _Z, _Z, _P]),
    "cfd_mse_loss_bwd_ld": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _Z, _Z, _Z, _P]),
    "cfd_rows_concat2": (_I, [_P, _Z, _Z, _P, _Z, _Z, _P, _Z, _P]),
    "cfd_loss_coef": (_I, [_P, _P, _I, _F, _P]),
    "cfd_label_energy_workspace_bytes": (_Z, []),
    "cfd_label_energy_coef": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "cfd_gelu_fwd": (_I, [_P, _P, _Z, _P]),
    "cfd_gelu_bwd": (_I, [_P, _P, _P, _Z, _P]),
    "cfd_adam_flat": (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _I, _F, _P]),
    "cfd_adam_multi": (_I, [_I, _P, _P, _P, _P, _P, _P, _F, _P, _F, _F, _F, _F, _F, _F, _P]),
    "cfd_scale_copy_multi": (_I, [_I, _P, _P, _P, _F, _P]),
    "cfd_gemm_workspace_bytes": (_Z, [_I, _I, _I]),
    "cfd_gemm": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cfd_linear_fwd_workspace_bytes": (_Z, [_I, _I, _I]),
    "cfd_linear_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfd_linear_bwd_workspace_bytes": (_Z, [_I, _I, _I]),
    "cfd_linear_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfd_linear_bwd_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "cfd_deeponet_inner_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfd_deeponet_inner_fwd_ex": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "cfd_deeponet_inner_bwd_workspace_bytes": (_Z, [_I, _I, _I]),
    "cfd_deeponet_inner_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfd_act_fwd": (_I, [_P, _P, _Z, _I, _P]),
    "cfd_act_bwd": (_I, [_P, _P, _P, _P, _Z, _I, _P]),
    "cfd_normact_fwd": (_I, [_P, _P, _P, _I, C.c_long, _I, _P]),
    "cfd_normact_bwd": (_I, [_P, _P, _P, _P, _I, C.c_long, _I, _P]),
    "cfd_bcast_add_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "cfd_bcast_add_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "cfd_rowdot_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "cfd_rowdot_bwd_workspace_bytes": (_Z, []),
    "cfd_rowdot_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfd_conv2d_fwd_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "cfd_conv2d_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cfd_conv2d_fwd_stats_slots": (_I, [_I, _I, _I, _I, _I, _I]),
    "cfd_conv2d_fwd_stats": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cfd_conv2d_bwd_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "cfd_conv2d_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cfd_conv2d_wfrag_bytes": (_Z, [_I, _I, _I, _I]),
    "cfd_conv2d_wprep_batch": (_I, [_I, _P, _P, _P, _P, _P, _P, _P]),
    "cfd_conv2d_fwd_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cfd_conv2d_bwd_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cfd_conv2d_zeropad_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "cfd_conv2d_zeropad_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cfd_conv2d_zeropad_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cfd_batchnorm_workspace_bytes": (_Z, [_I]),
    "cfd_batchnorm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _I, _P]),
    "cfd_batchnorm_fwd_stats": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _F, _F, _I, _P]),
    "cfd_batchnorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfd_maxpool2_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "cfd_maxpool2_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "cfd_maxpool2_bwd_add": (_I, [_P, _P, _P, _Z, _P, _I, _I, _I, _I, _P]),
    "cfd_upsample2_bilinear_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "cfd_upsample2_bilinear_bwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "cfd_convt2_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfd_convt2_bwd_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "cfd_convt2_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfd_convt2_fwd_ex": (_I, [_P, _P, _P, _P, C.c_long, _I, _I, _I, _I, _I, _P]),
    "cfd_convt2_bwd_ex": (_I, [_P, C.c_long, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfd_residual_mask": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfd_dropout": (_I, [_P, _P, _Z, _F, C.c_ulonglong, _P]),
    "cfd_dropout_gelu_fwd": (_I, [_P, _P, _Z, _F, C.c_ulonglong, _P]),
    "cfd_dropout_gelu_bwd": (_I, [_P, _P, _P, _Z, _F, C.c_ulonglong, _P]),
    "cfd_dropout_gelu_fwd_step": (_I, [_P, _P, _Z, _F, C.c_ulonglong, _P, _P]),
    "cfd_dropout_gelu_bwd_step": (_I, [_P, _P, _P, _Z, _F, C.c_ulonglong, _P, _P]),
    "cfd_ffn_stack_fwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "cfd_ffn_stack_bwd_workspace_bytes": (_Z, [_I, _P, _I]),
    "cfd_ffn_stacks_fwd": (_I, [_I, C.POINTER(FfnStackArgs), _P]),
    "cfd_ffn_stacks_bwd": (_I, [_I, C.POINTER(FfnStackArgs), _P]),
    "cfd_ffn_stack_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "cfd_fno_workspace_bytes": (_Z, [_P, C.POINTER(FnoShape), _I]),
    "cfd_fno_forward": (_I, [_P, C.POINTER(FnoShape), C.POINTER(FnoParams), _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "cfd_fno_workspace_bytes_ex": (_Z, [_P, C.POINTER(FnoShape), _I, _I]),
    "cfd_fno_forward_ex": (_I, [_P, C.POINTER(FnoShape), C.POINTER(FnoParams), _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "cfd_fno_forward_train": (_I, [_P, C.POINTER(FnoShape), C.POINTER(FnoParams), C.POINTER(FnoParams),
                                   _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P]),
    "cfd_fno_forward_train_ex": (_I, [_P, C.POINTER(FnoShape), C.POINTER(FnoParams), C.POINTER(FnoParams),
                                      _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _I, _P]),
    "cfd_fno_backward_phase_ex": (_I, [_P, C.POINTER(FnoShape), C.POINTER(FnoParams), C.POINTER(FnoParams),
                                       _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "cfd_fno_forward_train_f": (_I, [_P, C.POINTER(FnoShape), C.POINTER(FnoParams), C.POINTER(FnoParams),
                                     _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _I, _I, _P]),
    "cfd_fno_backward_phase_f": (_I, [_P, C.POINTER(FnoShape), C.POINTER(FnoParams), C.POINTER(FnoParams),
                                      _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfd_fno_adam_step": (_I, [_P, C.POINTER(FnoShape), C.POINTER(FnoParams), C.POINTER(FnoParams), _P, _P, _P, _P, _P,
                               _P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _I, _F, _I, _I, _I, _P]),
    "cfd_fno_backward": (_I, [_P, C.POINTER(FnoShape), C.POINTER(FnoParams), C.POINTER(FnoParams),
                              _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cfd_fno_backward_phase": (_I, [_P, C.POINTER(FnoShape), C.POINTER(FnoParams), C.POINTER(FnoParams),
                              _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
}


def exported_symbols():
    """Every symbol include/cfdbench_amd.h declares (checked against the built .so by the CPU tests)."""
    return sorted(_SIGS)


class CApi:
    def __init__(self, lib: C.CDLL, require_all: bool = True):
        self.lib = lib
        missing = []
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        if missing and require_all:
            raise CfdError(f"library {lib._name} lacks C-ABI symbols: {missing}")
        self.missing = missing
        if "cfd_version" not in missing and int(lib.cfd_version()) != ABI_VERSION:
            raise CfdError(f"library {lib._name} is ABI version {int(lib.cfd_version())}, this binding needs {ABI_VERSION} "
                           "(include/cfdbench_amd.h: CFD_ABI_VERSION) -- a stale build; run `python -m cfdbench_amd.build --force`")

    # -- helpers ---------------------------------------------------------------------------------
    def _chk(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.cfd_last_error()
            raise CfdError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")

    def call(self, name: str, *args):
        """Status-checked call of an ``int``-returning entry point."""
        self._chk(getattr(self.lib, name)(*args), name)

    def size(self, name: str, *args) -> int:
        return int(getattr(self.lib, name)(*args))

    def version(self) -> int:
        return int(self.lib.cfd_version())

    def plan_create(self, H: int, W: int, m1: int, m2: int) -> int:
        out = _P()
        self._chk(self.lib.cfd_plan_create(H, W, m1, m2, C.byref(out)), "cfd_plan_create")
        return out.value

    def plan_destroy(self, plan: Optional[int]):
        if plan:
            self.lib.cfd_plan_destroy(plan)
