"""Result-tree and checkpoint conventions of the reference (src/utils/common.py:16-33,161-284).  File names and JSON
schemas are what scripts/visualization/get_result.py:19-37 and plot_multistep_inference.py:107-110 read."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Optional, Union

import torch


def dump_json(data, path):  # common.py:16-23
    with open(path, "w", encoding="utf8") as f:
        json.dump(data, f, indent=4, ensure_ascii=False)


def load_json(path):  # common.py:26-33
    with open(path, "r", encoding="utf8") as f:
        return json.load(f)


def get_output_dir(args, is_auto: bool = False) -> Path:
    """result/{auto|non-auto}/{data_name}/dt{delta_time}/{model}/{hyper-parameter string}  (common.py:182-275)."""
    output_dir = Path(args.output_dir, "auto" if is_auto else "non-auto", args.data_name, f"dt{args.delta_time}", args.model)
    m = args.model
    if m == "deeponet":
        name = (f"lr{args.lr}_width{args.deeponet_width}_depthb{args.branch_depth}_deptht{args.trunk_depth}"
                f"_normprop{args.norm_props}_act{args.act_fn}-{args.act_scale_invariant}-{args.act_on_output}")
    elif m == "unet":
        name = f"lr{args.lr}_d{args.unet_dim}_cp{args.unet_insert_case_params_at}"
    elif m == "fno":
        name = f"lr{args.lr}_d{args.fno_depth}_h{args.fno_hidden_dim}_m1{args.fno_modes_x}_m2{args.fno_modes_y}"
    elif m == "resnet":
        name = f"lr{args.lr}_d{args.resnet_depth}_w{args.resnet_hidden_chan}"
    elif m == "auto_edeeponet":
        name = (f"lr{args.lr}_width{args.autoedeeponet_width}_depthb{args.autoedeeponet_depth}"
                f"_deptht{args.autoedeeponet_depth}_normprop{args.norm_props}_act{args.autoedeeponet_act_fn}")
    elif m == "auto_deeponet":
        name = (f"lr{args.lr}_width{args.deeponet_width}_depthb{args.branch_depth}_deptht{args.trunk_depth}"
                f"_normprop{args.norm_props}_act{args.act_fn}")
    elif m == "auto_ffn":
        name = f"lr{args.lr}_width{args.autoffn_width}_depth{args.autoffn_depth}"
    elif m == "auto_deeponet_cnn":
        name = f"lr{args.lr}_depth{args.autoffn_depth}"
    elif m == "ffn":
        name = f"lr{args.lr}_width{args.ffn_width}_depth{args.ffn_depth}"
    else:
        raise NotImplementedError(f"get_output_dir: model {m!r}")
    return output_dir / name


def get_best_ckpt(output_dir: Path) -> Union[Path, None]:
    """ckpt-* directory with the smallest dev_loss in its scores.json; None if there is none (common.py:161-174)."""
    best_loss, best = float("inf"), None
    for ckpt_dir in sorted(Path(output_dir).glob("ckpt-*")):
        dev_loss = load_json(ckpt_dir / "scores.json")["dev_loss"]
        if dev_loss < best_loss:
            best_loss, best = dev_loss, ckpt_dir
    return best


def load_ckpt(model, ckpt_path: Path) -> None:  # common.py:177-179
    print(f"Loading checkpoint from {ckpt_path}")
    model.load_state_dict(torch.load(ckpt_path, map_location="cpu"))


def load_best_ckpt(model, output_dir: Path) -> Path:  # common.py:278-284
    print(f"Finding the best checkpoint from {output_dir}")
    best = get_best_ckpt(output_dir)
    assert best is not None, f"no ckpt-*/scores.json under {output_dir}"
    print(f"Loading best checkpoint from {best}")
    load_ckpt(model, best / "model.pt")
    return best


# ---- plotting (common.py:35-158): same artefact names; out of scope for acceleration, skipped without matplotlib ----
def _plt():
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        return plt
    except Exception:  # noqa: BLE001
        return None


def plot_loss(losses, out: Path, fontsize: int = 12, linewidth: int = 2):
    plt = _plt()
    if plt is None:
        return
    plt.plot(losses, linewidth=linewidth)
    plt.xlabel("Step", fontsize=fontsize)
    plt.ylabel("Loss", fontsize=fontsize)
    plt.savefig(out)
    plt.clf()
    plt.close()


def plot_predictions(label, pred, out_dir: Path, step: int, inp: Optional[torch.Tensor] = None):
    """images/{input,label,pred}/{step:04d}.png as the reference writes them (common.py:35-93)."""
    plt = _plt()
    if plt is None:
        return
    for name, t in (("input", inp), ("label", label), ("pred", pred)):
        if t is None:
            continue
        d = Path(out_dir) / name
        d.mkdir(exist_ok=True, parents=True)
        plt.imsave(d / f"{step:04d}.png", t.detach().float().cpu().numpy(), cmap="coolwarm")


def plot(inp, label, pred, out_path: Path):
    plt = _plt()
    if plt is None:
        return
    fig, axs = plt.subplots(1, 3, figsize=(9, 3))
    for ax, t, title in zip(axs, (inp, label, pred), ("input", "label", "pred")):
        ax.imshow(t.detach().float().cpu().numpy(), cmap="coolwarm")
        ax.set_title(title)
        ax.axis("off")
    fig.savefig(out_path, bbox_inches="tight")
    plt.close(fig)
