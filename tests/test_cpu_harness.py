"""CPU: host logic of the harness (flags, result tree, collate, checkpoint selection, case sharding) against the
reference's conventions (src/args.py, src/utils/common.py, src/train_auto.py:33-58, src/test_multistep.py:85-92)."""
import json

import numpy as np
import pytest
import torch

from cfdbench_amd.harness.args import Args
from cfdbench_amd.harness.autoregressive import get_input_shapes
from cfdbench_amd.harness.common import dump_json, get_best_ckpt, get_output_dir
from cfdbench_amd.harness.data import SyntheticAutoDataset
from cfdbench_amd.harness.test_multistep import case_params_to_tensor
from cfdbench_amd.harness.train_auto import collate_fn


def test_flag_defaults_match_reference():
    a = Args()
    # src/args.py:22-217 (values read from the reference source)
    expect = dict(mode="train", seed=0, output_dir="result", lr=1e-4, weight_decay=1e-5, num_epochs=100, batch_size=8,
                  eval_batch_size=16, loss_name="mse", log_interval=50, eval_interval=2, data_name="cylinder_geo",
                  data_dir="../data", num_rows=64, num_cols=64, delta_time=0.1, norm_props=1, norm_bc=1,
                  model="pixel_diffusion", in_chan=2, out_chan=2, fno_depth=4, fno_hidden_dim=32, fno_modes_x=12,
                  fno_modes_y=12, unet_dim=12, unet_insert_case_params_at="input", resnet_depth=4, resnet_hidden_chan=16,
                  resnet_kernel_size=7, deeponet_width=100, branch_depth=8, trunk_depth=8, act_fn="relu",
                  act_scale_invariant=1, act_on_output=0, ffn_depth=8, ffn_width=100, autoffn_depth=8, autoffn_width=200)
    for k, v in expect.items():
        assert getattr(a, k) == v, k


def test_cli_parsing_data_alias_and_save(tmp_path):
    a = Args().parse_args(["--model", "fno", "--data", "cavity_prop_bc_geo", "--loss_name", "nmse", "--fno_hidden_dim", "20",
                           "--lr", "0.001"])
    assert a.data_name == "cavity_prop_bc_geo" and a.fno_hidden_dim == 20 and a.lr_step_size == 20
    p = tmp_path / "args.json"
    a.save(str(p))
    d = json.loads(p.read_text())
    assert d["model"] == "fno" and d["loss_name"] == "nmse"
    with pytest.raises(SystemExit):
        Args().parse_args(["--dat", "x"])  # no prefix matching: --data_name / --data_dir would be ambiguous


def test_is_args_valid_rejects_unimplemented_combinations():
    """src/args.py:372-378 plus this harness's own flags: nothing unimplemented is silently ignored."""
    from cfdbench_amd.harness.args import is_args_valid
    is_args_valid(Args(model="fno", data_name="cavity_prop_bc_geo"))
    is_args_valid(Args(model="fno", data_name="dam_prop", dtype="bf16"))
    is_args_valid(Args(model="fno", data_name="dam_prop", fused=1, lr_scheduler="cosine"))
    for bad in (dict(data_name="lid_driven"), dict(batch_size=0), dict(lr_scheduler="linear"), dict(dtype="fp16"),
                dict(model="unet", dtype="bf16"), dict(model="unet", fused=1), dict(model="fno", fused=1, gradient_accumulation_steps=2),
                dict(unet_insert_case_params_at="output"), dict(gradient_accumulation_steps=0)):
        kw = dict(model="fno", data_name="cavity_prop_bc_geo")
        kw.update(bad)
        with pytest.raises(AssertionError):
            is_args_valid(Args(**kw))


def test_output_dir_scheme():
    a = Args(model="fno", data_name="cavity_bc", lr=0.001, fno_hidden_dim=20)
    assert str(get_output_dir(a, is_auto=True)) == "result/auto/cavity_bc/dt0.1/fno/lr0.001_d4_h20_m112_m212"
    a = Args(model="unet", data_name="cylinder_prop_bc_geo")
    assert str(get_output_dir(a, is_auto=True)) == "result/auto/cylinder_prop_bc_geo/dt0.1/unet/lr0.0001_d12_cpinput"
    a = Args(model="deeponet", data_name="tube_prop_geo")
    assert str(get_output_dir(a, is_auto=False)).endswith(
        "non-auto/tube_prop_geo/dt0.1/deeponet/lr0.0001_width100_depthb8_deptht8_normprop1_actrelu-1-0")
    a = Args(model="auto_deeponet", data_name="tube_prop_geo")
    assert str(get_output_dir(a, True)).endswith("auto_deeponet/lr0.0001_width100_depthb8_deptht8_normprop1_actrelu")


def test_input_shapes():
    assert get_input_shapes(Args(data_name="cavity_prop_bc_geo")) == (64, 64, 5)
    assert get_input_shapes(Args(data_name="dam_prop_bc_geo")) == (66, 65, 5)
    assert get_input_shapes(Args(data_name="cylinder_geo")) == (66, 65, 8)


def test_collate_splits_mask_and_orders_case_params():
    ds = SyntheticAutoDataset(n_cases=2, n_frames=3, height=8, width=8, seed=1)
    items = [ds[0], ds[3]]
    # case.json may carry rotated/dx/dy, which the reference drops (train_auto.py:44-47)
    items[0] = (items[0][0], items[0][1], dict(items[0][2], rotated=1, dx=0.1, dy=0.2))
    items[1] = (items[1][0], items[1][1], dict(items[1][2], rotated=0, dx=0.1, dy=0.2))
    b = collate_fn(items, device=None)
    assert b["inputs"].shape == (2, 2, 8, 8) and b["label"].shape == (2, 2, 8, 8) and b["mask"].shape == (2, 1, 8, 8)
    assert b["case_params"].shape == (2, 5)
    np.testing.assert_allclose(b["case_params"][0].numpy(),
                               [items[0][2][k] for k in ("vel_top", "density", "viscosity", "height", "width")], rtol=1e-6)
    assert torch.equal(b["mask"][0, 0], items[0][0][-1]) and torch.equal(b["label"][1], items[1][1][:-1])
    t = case_params_to_tensor(items[0][2])
    assert torch.allclose(t, b["case_params"][0])


def test_best_checkpoint_selection(tmp_path):
    assert get_best_ckpt(tmp_path) is None
    for ep, loss in ((1, 0.5), (3, 0.2), (5, 0.3)):
        d = tmp_path / f"ckpt-{ep}"
        d.mkdir()
        dump_json(dict(ep=ep, train_loss=1.0, dev_loss=loss, time=1.0), d / "scores.json")
    assert get_best_ckpt(tmp_path).name == "ckpt-3"


def test_nonauto_collate_and_init_model_shapes():
    """src/train.py:26-35,256-291: batch keys, and the branch / trunk widths init_model derives from the flags."""
    from cfdbench_amd.harness.train import SyntheticDataset, collate_fn as collate_nonauto, init_model as init_nonauto
    ds = SyntheticDataset(2, 3, 8, 8, seed=0)
    b = collate_nonauto([ds[0], ds[4]], device=None)
    assert b["case_params"].shape == (2, 5) and b["t"].shape == (2, 1) and b["label"].shape == (2, 3, 8, 8)
    assert float(b["t"][1, 0]) == 1.0
    m = init_nonauto(Args(model="deeponet", data_name="cylinder_geo", deeponet_width=16, branch_depth=2, trunk_depth=3))
    assert m.branch_net.fc_layers if hasattr(m.branch_net, "fc_layers") else True
    sd = m.state_dict()
    assert sd["fc_trunk_t.weight"].shape[1] == 1 and sd["fc_trunk_xy.weight"].shape[1] == 2
    assert [v.shape[1] for k, v in sd.items() if k.startswith("branch_net") and k.endswith("weight")][0] == 8
    f = init_nonauto(Args(model="ffn", data_name="cavity_bc", ffn_width=12, ffn_depth=2))
    ws = [tuple(v.shape) for k, v in f.state_dict().items() if k.endswith("weight")]
    assert ws == [(12, 8), (12, 12), (1, 12)]
    with pytest.raises(ValueError):
        init_nonauto(Args(model="fno", data_name="cavity_bc"))


def test_init_model_resolves_every_autoregressive_name():
    """Same ``--model`` names as src/utils/autoregressive.py:41-125 (construction only: no kernel runs on the CPU)."""
    from cfdbench_amd.harness.autoregressive import init_model
    names = {"fno": "Fno2d", "unet": "UNet", "resnet": "ResNet", "auto_deeponet": "AutoDeepONet",
             "auto_edeeponet": "AutoEDeepONet", "auto_ffn": "AutoFfn", "auto_deeponet_cnn": "AutoDeepONetCnn"}
    for name, cls in names.items():
        m = init_model(Args(model=name, data_name="cavity_prop_bc_geo", num_rows=16, num_cols=16, deeponet_width=8,
                            autoffn_width=8, autoedeeponet_width=8, fno_hidden_dim=4, unet_dim=2, resnet_hidden_chan=2))
        assert type(m).__name__ == cls and hasattr(m, "generate_many")
    with pytest.raises(ValueError):
        init_model(Args(model="no_such_model", data_name="cavity_bc"))


def test_device_batch_loader_equals_dataloader_plus_collate():
    """batches gathered from the stacked frames == the reference's item-by-item collate (same order)."""
    from functools import partial
    from torch.utils.data import DataLoader
    from cfdbench_amd.harness.data import DeviceBatchLoader
    ds = SyntheticAutoDataset(n_cases=3, n_frames=5, height=8, width=9, seed=4, border_mask=True)
    ref = list(DataLoader(ds, batch_size=5, shuffle=False, collate_fn=partial(collate_fn, device=None)))
    got = list(DeviceBatchLoader(ds, 5, shuffle=False, device="cpu"))
    assert len(ref) == len(got) == len(DeviceBatchLoader(ds, 5, shuffle=False, device="cpu")) == 3
    for a, b in zip(ref, got):
        assert set(a) == set(b) == {"inputs", "label", "mask", "case_params"}
        for k in a:
            assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
            assert b[k].is_contiguous()
    # a shard (data-parallel subset) in shuffled order: every frame of the shard exactly once, drop_last honoured
    idx = [11, 2, 7, 5, 0, 9, 3]
    torch.manual_seed(5)
    ld = DeviceBatchLoader(ds, 3, shuffle=True, drop_last=True, device="cpu", indices=idx)
    batches = list(ld)
    assert len(batches) == len(ld) == 2
    seen = torch.cat([b["inputs"] for b in batches])
    pool = ds.inputs[idx][:, :-1]
    assert all(any(torch.equal(x, p) for p in pool) for x in seen) and len(seen) == 6
    torch.manual_seed(5)
    again = torch.cat([b["inputs"] for b in DeviceBatchLoader(ds, 3, shuffle=True, drop_last=True, device="cpu", indices=idx)])
    assert torch.equal(seen, again)  # the permutation comes from the host generator: reproducible / resumable


def test_device_batch_loader_gathers_into_bound_buffers():
    """DeviceBatchLoader.bind (round 6): full-size batches are gathered straight into the buffers of a captured training step
    (GraphedTrainStep.static) -- the yielded tensors ARE those buffers, with the same values an unbound loader yields; a short last batch
    comes as fresh tensors."""
    from cfdbench_amd.harness.data import DeviceBatchLoader
    ds = SyntheticAutoDataset(n_cases=3, n_frames=5, height=8, width=9, seed=4, border_mask=True)  # 12 frames: batches of 5, 5, 2
    plain = list(DeviceBatchLoader(ds, 5, shuffle=False, device="cpu"))
    ld = DeviceBatchLoader(ds, 5, shuffle=False, device="cpu")
    static = {k: torch.full_like(v, -1.0) for k, v in plain[0].items()}
    ld.bind(static)
    for i, b in enumerate(ld):
        for k in b:
            assert torch.equal(b[k], plain[i][k]), (i, k)
            assert (b[k].data_ptr() == static[k].data_ptr()) == (i < 2), (i, k)  # the short last batch is not the bound buffer
    with pytest.raises(ValueError):
        ld.bind({k: v[:, :0] if v.dim() > 1 else v for k, v in static.items()})
    ld.bind(None)
    assert next(iter(ld))["inputs"].data_ptr() != static["inputs"].data_ptr()


def test_lr_schedules_follow_torch_and_early_stopping():
    """harness/schedule.py: the fused engine's shadow-optimizer schedule == the torch scheduler on a real optimizer."""
    import torch
    from cfdbench_amd.harness.schedule import EarlyStopping, LrSchedule
    for kind in ("step", "cosine", "plateau"):
        opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(3))], lr=2e-3)
        a = LrSchedule(kind, 2e-3, 10, opt, lr_step_size=3, lr_gamma=0.9, factor=0.5, patience=1)
        b = LrSchedule(kind, 2e-3, 10, None, lr_step_size=3, lr_gamma=0.9, factor=0.5, patience=1)
        dev = [1.0, 0.9, 0.95, 0.97, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]
        seen = []
        for ep in range(10):
            assert a.lr == b.lr
            seen.append(a.lr)
            for s_ in (a, b):
                s_.epoch_end()
                s_.validation(dev[ep])
        if kind == "step":
            assert abs(seen[3] - 2e-3 * 0.9) < 1e-15 and abs(seen[9] - 2e-3 * 0.9 ** 3) < 1e-15
        if kind == "cosine":
            assert abs(seen[5] - 1e-3) < 1e-12
        if kind == "plateau":
            assert seen[0] == seen[2] == 2e-3 and seen[4] == 1e-3 and seen[-1] < 1e-3
        st = b.state_dict()
        c = LrSchedule(kind, 2e-3, 10, None, lr_step_size=3, lr_gamma=0.9, factor=0.5, patience=1)
        c.load_state_dict(st)
        assert c.lr == b.lr
    es = EarlyStopping(patience=2, delta=0.01)
    assert [es.update(v) for v in (1.0, 0.995, 0.5, 0.495, 0.499)] == [False, False, False, False, True]
    assert not EarlyStopping(0).update(1.0)
