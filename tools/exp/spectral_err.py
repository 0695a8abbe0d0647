#!/usr/bin/env python
"""GPU box: measured error of the contraction kernels against the fp64 oracle (relative nMSE), for the arithmetic record in
DESIGN.md: SpectralConv2d forward / backward (64x64 and 66x65), the fused FnoBlock, the 1x1 conv / weight gradient, the whole model."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import kernel_checks as K  # noqa: E402
from backends import TorchBackend  # noqa: E402

be = TorchBackend()
out = {}
out["spectral_64x64_B8_C20"] = K.check_spectral(be, 8, 20, 20, 64, 64)
out["spectral_66x65_B4_C32"] = K.check_spectral(be, 4, 32, 32, 66, 65)
out["block_64x64_B4_C20"] = K.check_block(be, 4, 20, 20, 64, 64)
out["block_66x65_B2_C32"] = K.check_block(be, 2, 32, 32, 66, 65)
out["chanmix_C20_act"] = K.check_chanmix(be, 4, 20, 20, 4096, 1)
out["fno_model_B4_C20_L4"] = K.check_fno_vs_oracle(be, 4, 20, 4, 64, 64)
print(json.dumps({k: ({kk: float(f"{vv:.3e}") for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in out.items()}, indent=1))
