"""``nr3d_lib.checkpoint.CheckpointIO`` as the trainer drives it (code_single/tools/train.py:1363-1385, 1655-1689):
``register_modules(**name -> module | optimizer)``, ``save(filename=, **scalars)``, ``load_file(path | None,
ignore_keys=, only_use_keys=, map_location=) -> dict of the saved scalars``; ``sorted_ckpts(dir)``."""
import os
from typing import List

import torch


def sorted_ckpts(checkpoint_dir: str) -> List[str]:
    if not os.path.isdir(checkpoint_dir):
        return []
    f = [x for x in os.listdir(checkpoint_dir) if x.endswith(".pt")]
    # numbered checkpoints, then ``latest.pt`` (rewritten every ``i_save`` seconds DURING training), then ``final_*.pt``
    # (written once, after the last iteration -- train.py:1684): ``sorted_ckpts(d)[-1]`` of a finished run is its final
    # state, of an interrupted run the most recent ``latest.pt`` (render.py:59, extract_mesh.py:38 and the resume path
    # all take ``[-1]``)
    key = lambda n: (n.startswith("final"), n.startswith("latest"), n)       # noqa: E731
    return [os.path.join(checkpoint_dir, x) for x in sorted(f, key=key)]


class CheckpointIO:
    def __init__(self, checkpoint_dir: str = "./chkpts", allow_mkdir: bool = True, **modules):
        self.checkpoint_dir, self.modules = checkpoint_dir, dict(modules)
        if allow_mkdir:
            os.makedirs(checkpoint_dir, exist_ok=True)

    def register_modules(self, **modules):
        self.modules.update(modules)

    def save(self, filename: str, **scalars):
        path = filename if os.path.isabs(filename) else os.path.join(self.checkpoint_dir, filename)
        out = dict(scalars)
        for k, m in self.modules.items():
            out[k] = m.state_dict()
        torch.save(out, path)
        return path

    def load_file(self, filepath=None, ignore_keys=None, only_use_keys=None, map_location="cpu") -> dict:
        if filepath is None:
            cks = sorted_ckpts(self.checkpoint_dir)
            if not cks:
                return dict()
            filepath = cks[-1]
        state = torch.load(filepath, map_location=map_location, weights_only=False)
        return self.parse_state_dict(state, ignore_keys=ignore_keys, only_use_keys=only_use_keys)

    load = load_file

    def parse_state_dict(self, state: dict, ignore_keys=None, only_use_keys=None) -> dict:
        ignore_keys = list(ignore_keys or [])
        scalars = {}
        for k, v in state.items():
            if k in self.modules:
                if k in ignore_keys or (only_use_keys and k not in only_use_keys):
                    continue
                self.modules[k].load_state_dict(v)
            else:
                scalars[k] = v
        return scalars
