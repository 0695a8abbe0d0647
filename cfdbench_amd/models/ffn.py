"""``Ffn``: the Linear/activation stack every DeepONet variant is built from (src/models/ffn.py:12-35), with the
reference's module layout (``layers.{0,2,4,...}.{weight,bias}`` in the state_dict) and every Linear+activation pair
running as ONE MFMA GEMM with a fused bias/activation epilogue."""
from torch import Tensor, nn

from .. import functional as F_
from .act_fn import _Act


class Ffn(nn.Module):
    def __init__(self, dims: list, act_fn: nn.Module, act_on_output: bool = False):
        super().__init__()
        self.dims = dims
        layers = []
        for i in range(len(dims) - 2):
            layers.append(nn.Linear(dims[i], dims[i + 1]))
            layers.append(act_fn)
        layers.append(nn.Linear(dims[-2], dims[-1]))
        if act_on_output:
            layers.append(act_fn)
        self.layers = nn.Sequential(*layers)  # parameter container only; forward() below fuses pairs

    def forward(self, x: Tensor) -> Tensor:
        mods = list(self.layers)
        i = 0
        while i < len(mods):
            lin = mods[i]
            act = None
            if i + 1 < len(mods) and isinstance(mods[i + 1], _Act):
                act = mods[i + 1].name
                i += 1
            x = F_.linear_act(x, lin.weight, lin.bias, act)
            i += 1
        return x
