"""``Ffn``: the Linear/activation stack every DeepONet variant is built from (src/models/ffn.py:12-35), with the
reference's module layout (``layers.{0,2,4,...}.{weight,bias}`` in the state_dict) and every Linear+activation pair
running as ONE MFMA GEMM with a fused bias/activation epilogue."""
from torch import Tensor, nn

from .. import functional as F_
from .act_fn import NormAct, _Act
from .base_model import CfdModel


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
        return self._run(x, 0)

    def forward_from(self, z: Tensor, idx: int) -> Tensor:
        """Continue after Linear ``layers[idx]`` whose output ``z`` the caller computed itself (its activation, if any,
        is applied here as a stand-alone kernel)."""
        mods = list(self.layers)
        i = idx + 1
        if i < len(mods) and isinstance(mods[i], _Act):
            z = F_.act(z, mods[i].name)
            i += 1
        elif i < len(mods) and isinstance(mods[i], NormAct):
            z = mods[i](z)
            i += 1
        return self._run(z, i)

    def _fusable_run(self, mods, i: int):
        """(weights, biases, act name, act after the last layer, index behind the run) of the longest run of Linear[+plain
        activation] layers from ``mods[i]`` on whose widths are all <= 128 -- the stack one kernel runs per direction
        (functional.FfnStackFn, csrc/ffn.hip) -- or None when fewer than two layers qualify."""
        ws, bs, act, j, last_act = [], [], None, i, False
        while j < len(mods) and isinstance(mods[j], nn.Linear) and len(ws) < F_.FFN_STACK_MAX_LAYERS:
            lin = mods[j]
            if max(lin.in_features, lin.out_features) > F_.FFN_STACK_MAX_WIDTH:
                break
            nxt = mods[j + 1] if j + 1 < len(mods) else None
            if isinstance(nxt, NormAct):
                break  # statistics over the whole sample: not a per-row epilogue
            if isinstance(nxt, _Act):
                if act is not None and nxt.name != act:
                    break
                if ws and not last_act:
                    break  # a layer without activation in the middle of the run ends it
                act = nxt.name
                ws.append(lin.weight); bs.append(lin.bias); last_act = True
                j += 2
            else:
                if ws and not last_act:
                    break
                ws.append(lin.weight); bs.append(lin.bias); last_act = False
                j += 1
                break  # a Linear without activation can only close a run
        return (ws, bs, act, last_act, j) if len(ws) >= 2 else None

    def _chain_run(self, mods, i: int):
        """(weights, biases, activation name per layer, index behind the run) of the longest run of Linear[+plain activation] layers
        from ``mods[i]`` on (any width; a NormAct ends it: its statistics span the whole sample; so does the start of a run the stack
        kernel takes) -- one autograd node whose backward
        pass folds each activation derivative into the GEMM above it (functional.LinearChainFn) -- or None for fewer than two."""
        ws, bs, acts, j = [], [], [], i
        while j < len(mods) and isinstance(mods[j], nn.Linear):
            if j > i and self._fusable_run(mods, j) is not None:
                break  # the narrow layers behind a wide first one belong to the stack kernel
            nxt = mods[j + 1] if j + 1 < len(mods) else None
            if isinstance(nxt, NormAct):
                break
            ws.append(mods[j].weight); bs.append(mods[j].bias)
            if isinstance(nxt, _Act):
                acts.append(nxt.name)
                j += 2
            else:
                acts.append(None)
                j += 1
        return (ws, bs, acts, j) if len(ws) >= 2 else None

    def _run(self, x: Tensor, i: int, stop_at_fusable: bool = False):
        """Layers from ``mods[i]`` on: stack kernel / chain node / single Linear(+activation), in that order of preference.
        ``stop_at_fusable``: return ``(x, i)`` as soon as ``mods[i]`` starts a run the stack kernel takes (run_ffns_together lets
        the runs of several nets share one launch) -- ONE routing for both callers."""
        mods = list(self.layers)
        while i < len(mods):
            run = self._fusable_run(mods, i) if x.is_cuda else None
            if run is not None and stop_at_fusable:
                return x, i
            if run is not None:
                ws, bs, act, last_act, i = run
                x = F_.ffn_stack(x, ws, bs, act, last_act)
                continue
            chain = self._chain_run(mods, i) if x.is_cuda else None
            if chain is not None:
                ws, bs, acts, i = chain
                x = F_.linear_chain(x, ws, bs, acts)
                continue
            lin = mods[i]
            act = None
            norm = None
            if i + 1 < len(mods) and isinstance(mods[i + 1], _Act):
                act = mods[i + 1].name
                i += 1
            elif i + 1 < len(mods) and isinstance(mods[i + 1], NormAct):
                norm = mods[i + 1]  # statistics span the whole sample, so the activation cannot ride in the GEMM epilogue
                i += 1
            x = F_.linear_act(x, lin.weight, lin.bias, act)
            if norm is not None:
                x = norm(x)
            i += 1
        return (x, i) if stop_at_fusable else x


def run_ffns_together(ffns, xs):
    """``[ffn(x) for ffn, x in zip(ffns, xs)]`` for INDEPENDENT nets (a DeepONet's branch and trunk), with the fusable runs of different
    nets sharing one kernel launch per direction (functional.ffn_stacks): each net first runs the layers that no stack kernel takes
    (a first Linear wider than 128), then the nets that stand at a fusable run go together, then each finishes on its own."""
    states = []  # [net, module index, tensor]
    for net, x in zip(ffns, xs):
        x, i = net._run(x, 0, stop_at_fusable=True)  # the prefix no stack kernel takes (a chain node or single layers: Ffn._run's routing)
        states.append([net, i, x])
    ready = [k for k, (net, i, x) in enumerate(states) if x.is_cuda and i < len(list(net.layers))]
    if 2 <= len(ready) <= F_.FFN_STACKS_MAX:
        runs = []
        for k in ready:
            net, i, x = states[k]
            ws, bs, act, last_act, j = net._fusable_run(list(net.layers), i)
            runs.append((x, ws, bs, act, last_act))
            states[k][1] = j
        for k, y in zip(ready, F_.ffn_stacks(runs)):
            states[k][2] = y
    return [net._run(x, i) for net, i, x in states]


class FfnModel(CfdModel):
    """Non-autoregressive FFN baseline (src/models/ffn.py:38-181): one Ffn over [case params, x, y, t] rows."""

    def __init__(self, loss_fn, widths, act_name: str = "relu", act_norm: bool = True, act_on_output: bool = False,
                 num_label_samples: int = 1000):
        super().__init__(loss_fn)
        from .act_fn import get_act_fn
        self.loss_fn = loss_fn
        self.widths = widths
        self.act_name = act_name
        self.act_norm = act_norm
        self.act_on_output = act_on_output
        self.num_label_samples = num_label_samples
        self.ffn = Ffn(self.widths, act_fn=get_act_fn(act_name, act_norm), act_on_output=self.act_on_output)

    def forward(self, case_params: Tensor, t: Tensor, label=None, query_idxs=None):
        """case_params (b,p), t (b,1), label (b,c,h,w), query_idxs (k,2) -> preds (b,k) [+ loss]  (ffn.py:73-146)."""
        import torch
        batch_size, dim_in = case_params.shape
        if query_idxs is None:
            assert label is not None
            height, width = label.shape[-2:]
            query_idxs = torch.stack([torch.randint(0, height, (self.num_label_samples,), device=label.device),
                                      torch.randint(0, width, (self.num_label_samples,), device=label.device)], dim=-1)
        coords = query_idxs.unsqueeze(0).repeat(batch_size, 1, 1)
        num_queries = coords.shape[1]
        tt = t.unsqueeze(-1).repeat(1, num_queries, 1)
        coords = torch.cat([coords, tt], dim=-1)
        cp = case_params.unsqueeze(1).repeat(1, num_queries, 1)
        inp = torch.cat([cp, coords], dim=-1).view(batch_size * num_queries, -1)
        preds = self.ffn(inp).view(batch_size, num_queries)
        if label is not None:
            labels = label[:, 0][:, query_idxs[:, 0], query_idxs[:, 1]]
            assert preds.shape == labels.shape, f"{preds.shape}, {labels.shape}"
            return dict(preds=preds, loss=self.loss_fn(preds=preds, labels=labels))
        return dict(preds=preds)

    def generate_one(self, case_params: Tensor, t: Tensor, height: int, width: int) -> Tensor:
        import torch
        from itertools import product
        if len(case_params.shape) == 1:
            case_params = case_params.unsqueeze(0)
        if len(t.shape) == 0:
            t = t.unsqueeze(0).unsqueeze(0)
        elif len(t.shape) == 1:
            t = t.unsqueeze(0)
        query_idxs = torch.tensor(list(product(range(height), range(width))), device=case_params.device)
        return self.forward(case_params, t=t, query_idxs=query_idxs)["preds"].view(-1, 1, height, width)
