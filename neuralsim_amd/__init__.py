"""neuralsim_amd -- the NeuS / StreetSurf render-and-train hot path on MI355X.

Importing this package changes no process-wide state.  One process drives one GPU (torch.distributed over RCCL across
processes): the autograd engine's per-device worker thread has nothing to run in parallel with, and handing every backward
over to it costs a thread switch plus GIL ping-pong between the two threads for each of the Python-side backward functions
(``fields/neus.py``) -- measured on the drop-in API path of the headline step (renderer + autograd functions,
``NSIM_FUSED_STEP=0``): 2.23 -> 1.95 ms per step (p50) with the backward on the calling thread.  That is therefore how THIS
package's own trainers run their backward -- scoped with ``backward_on_calling_thread()`` (round 5; rounds 3-4 switched it off
for the whole process at import, which took the per-device backward threads away from unrelated code in the same process).
The reference's trainer calls ``loss.backward()`` itself: the process that owns it decides (tools/run_reference_train.py
switches it process-wide before it hands over to train.py).  ``NSIM_AUTOGRAD_MT=1`` keeps torch's default."""
import contextlib
import os

import torch


def backward_on_calling_thread():
    """context manager: ``loss.backward()`` inside it runs on the calling thread (restored afterwards)."""
    if os.environ.get("NSIM_AUTOGRAD_MT", "0") == "1":
        return contextlib.nullcontext()
    return torch.autograd.set_multithreading_enabled(False)
