"""``nr3d_lib.models.blocks`` -- plain MLP blocks (docs/exps/exp_lipshitz_3d.py:46-49: ``MLP(in, out, D=, W=, dtype=,
device=)``, D = number of hidden layers; app/models/env/sky.py:27 ``get_blocks``).  The hot-path decoders of this
repository are the fused HIP kernels; these torch blocks exist for the harness-side models only."""
import torch
import torch.nn as nn

_ACT = dict(relu=nn.ReLU, softplus=nn.Softplus, sigmoid=nn.Sigmoid, tanh=nn.Tanh, none=nn.Identity, identity=nn.Identity)


def _act(a):
    if a is None:
        return nn.Identity()
    if isinstance(a, dict):
        kw = {k: v for k, v in a.items() if k != "type"}
        return _ACT[a["type"]](**kw)
    return _ACT[str(a)]()


class MLP(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, D: int = 2, W: int = 64, activation="relu", output_activation=None,
                 dtype=torch.float32, device=None, **unused):
        super().__init__()
        dims = [in_ch] + [W] * D + [out_ch]
        layers = []
        for i in range(len(dims) - 1):
            layers.append(nn.Linear(dims[i], dims[i + 1], dtype=torch.float32, device=device))
            layers.append(_act(activation) if i < len(dims) - 2 else _act(output_activation))
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        return self.layers(x.float())


LipshitzMLP = MLP


def get_blocks(in_ch: int, out_ch: int, type: str = "mlp", **kw):
    return MLP(in_ch, out_ch, **{k: v for k, v in kw.items() if k not in ("use_tcnn_backend",)})
