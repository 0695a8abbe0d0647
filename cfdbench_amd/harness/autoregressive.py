"""``get_input_shapes`` / ``init_model``: the reference's plugin registration point (src/utils/autoregressive.py:19-125)."""
from __future__ import annotations

from typing import Tuple

from ..models.auto_deeponet import AutoDeepONet
from ..models.auto_deeponet_cnn import AutoDeepONetCnn
from ..models.auto_edeeponet import AutoEDeepONet
from ..models.auto_ffn import AutoFfn
from ..models.base_model import AutoCfdModel
from ..models.fno.fno2d import Fno2d
from ..models.loss import loss_name_to_fn
from ..models.resnet import ResNet
from ..models.unet import UNet


def get_input_shapes(args) -> Tuple[int, int, int]:
    """(rows, cols, n_case_params) from data_name / num_rows / num_cols (autoregressive.py:19-38)."""
    if any(x in args.data_name for x in ["tube", "dam", "cylinder"]):
        n_rows, n_cols = args.num_rows + 2, args.num_cols + 1  # top/bottom and left boundaries
    else:
        assert "cavity" in args.data_name
        n_rows, n_cols = args.num_rows, args.num_cols
    if "cylinder" in args.data_name:
        n_case_params = 8  # vel_in, density, viscosity, height, width, radius, center_x, center_y
    else:
        assert any(x in args.data_name for x in ["cavity", "tube", "dam"])
        n_case_params = 5
    return n_rows, n_cols, n_case_params


def init_model(args) -> AutoCfdModel:
    """Same ``elif`` chain as autoregressive.py:41-179; models whose kernels are not built yet name themselves."""
    loss_fn = loss_name_to_fn(args.loss_name)
    n_rows, n_cols, n_case_params = get_input_shapes(args)
    if args.model == "fno":
        return Fno2d(in_chan=args.in_chan, out_chan=args.out_chan, n_case_params=n_case_params, loss_fn=loss_fn,
                     num_layers=args.fno_depth, hidden_dim=args.fno_hidden_dim, modes1=args.fno_modes_x,
                     modes2=args.fno_modes_y)
    if args.model == "auto_deeponet":  # autoregressive.py:58-69
        return AutoDeepONet(branch_dim=n_cols * n_rows + n_case_params, trunk_dim=2, loss_fn=loss_fn,
                            width=args.deeponet_width, trunk_depth=args.trunk_depth, branch_depth=args.branch_depth,
                            act_name=args.act_fn)
    if args.model == "unet":  # autoregressive.py:105-114
        return UNet(in_chan=args.in_chan, out_chan=args.out_chan, loss_fn=loss_fn, n_case_params=n_case_params,
                    insert_case_params_at=args.unet_insert_case_params_at, dim=args.unet_dim)
    if args.model == "resnet":  # autoregressive.py:93-104
        return ResNet(in_chan=args.in_chan, out_chan=args.out_chan, n_case_params=n_case_params, loss_fn=loss_fn,
                      hidden_chan=args.resnet_hidden_chan, num_blocks=args.resnet_depth,
                      kernel_size=args.resnet_kernel_size, padding=args.resnet_padding)
    if args.model == "auto_ffn":  # autoregressive.py:48-57
        return AutoFfn(input_field_dim=n_rows * n_cols, num_case_params=n_case_params, query_dim=2, loss_fn=loss_fn,
                       width=args.autoffn_width, depth=args.autoffn_depth)
    if args.model == "auto_edeeponet":  # autoregressive.py:70-81
        return AutoEDeepONet(dim_branch1=n_rows * n_cols, dim_branch2=n_case_params, trunk_dim=2, loss_fn=loss_fn,
                             width=args.autoedeeponet_width, trunk_depth=args.autoedeeponet_depth,
                             branch_depth=args.autoedeeponet_depth, act_name=args.autoedeeponet_act_fn)
    if args.model == "auto_deeponet_cnn":  # autoregressive.py:82-91
        return AutoDeepONetCnn(in_chan=args.in_chan, height=n_rows, width=n_cols, num_case_params=n_case_params,
                               query_dim=2, loss_fn=loss_fn)
    raise ValueError(f"Invalid model name: {args.model}")
