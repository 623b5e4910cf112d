"""neuralsim_amd -- the NeuS / StreetSurf render-and-train hot path on MI355X.

One process drives one GPU (torch.distributed over RCCL across processes): the autograd engine's per-device worker thread has
nothing to run in parallel with, and handing every backward over to it costs a thread switch plus GIL ping-pong between the
two threads for each of the Python-side backward functions (``fields/neus.py``) -- measured on the drop-in API path of the
headline step (renderer + autograd functions, ``NSIM_FUSED_STEP=0``): 2.23 -> 1.95 ms per step (p50) with the backward on
the calling thread.  ``NSIM_AUTOGRAD_MT=1`` keeps torch's default."""
import os

import torch

if os.environ.get("NSIM_AUTOGRAD_MT", "0") != "1":
    torch.autograd.set_multithreading_enabled(False)
