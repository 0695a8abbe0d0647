"""Activation selection with the reference's names (src/models/act_fn.py:5-18).  The returned modules are markers:
``Ffn`` fuses the activation into the preceding Linear's GEMM epilogue (functional.linear_act), so they are never
called on their own."""
from torch import nn


class _Act(nn.Module):
    name = "none"

    def forward(self, x):
        raise RuntimeError(f"{type(self).__name__} is fused into the preceding Linear inside cfdbench_amd.Ffn; "
                           "it is not a stand-alone op")


class ReLU(_Act):
    name = "relu"


class Tanh(_Act):
    name = "tanh"


class GELU(_Act):
    name = "gelu"


class SiLU(_Act):
    name = "swish"


def get_act_fn(name: str, norm: bool = False) -> nn.Module:
    fns = {"relu": ReLU, "tanh": Tanh, "gelu": GELU, "swish": SiLU}
    if name not in fns:
        raise ValueError(f"Unknown activation function: {name}")
    fn = fns[name]()
    return NormAct(fn) if norm else fn


class NormAct(nn.Module):
    """Normalised activation (act_fn.py:21-47): per sample, normalise -> act -> transform back.  Marker for ``Ffn``
    (which runs the preceding Linear without activation and then the fused NormAct kernel); also callable on its own."""

    def __init__(self, act_fn: nn.Module):
        super().__init__()
        self.act_fn = act_fn

    def forward(self, x):
        from .. import functional as F_
        return F_.NormActFn.apply(x, F_.ACT_CODES[self.act_fn.name])
