"""``nr3d_lib.distributed`` (reference import: code_single/tools/train.py:38, used :1195-1204): ``init_env(args, seed=)``
seeds the process and -- under ``args.ddp`` / torchrun -- joins the RCCL process group; rank helpers."""
import os
import random

import numpy as np
import torch

from neuralsim_amd import distributed as _nd
from neuralsim_amd.distributed import get_rank, get_world_size, is_master  # noqa: F401


def get_local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def init_env(args=None, seed: int = 42, backend: str = None, device_type: str = None, **unused):
    """-> (rank, local_rank, world_size).  ``args`` is the trainer's config (``args.ddp``); called without it by this
    repository's own code with (backend=, device_type=)."""
    if device_type is None:
        device_type = "cuda" if torch.cuda.is_available() else "cpu"
    ddp = bool(args.get("ddp", False)) if args is not None else int(os.environ.get("WORLD_SIZE", "1")) > 1
    if ddp or args is None:
        rank, local_rank, world = _nd.init_env(backend=backend, device_type=device_type)
        if args is not None and ddp:
            # one process per GPU: the trainer builds ``torch.device(f'cuda:{device_ids[0]}')`` (train.py:77) and hands
            # ``device_ids`` to DistributedDataParallel (:1405) -- both must name THIS rank's device
            args["device_ids"] = [local_rank]
    else:
        rank, local_rank, world = 0, 0, 1
    s = int(seed) + rank
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)
    return rank, local_rank, world
