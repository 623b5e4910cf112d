"""``nr3d_lib.profile.profile`` as the renderers use it (``@profile`` on methods, ``with profile("name"):`` around
phases -- app/renderers/single_volume_renderer.py:15,136,235): a no-op here.  Timings of this path come from
rocprofv3 / HIP events (bench.py), not from a Python-side profiler."""
import contextlib
import functools


class _Profile:
    enabled = False

    def __call__(self, arg=None):
        if callable(arg):                       # @profile
            @functools.wraps(arg)
            def wrapped(*a, **k):
                return arg(*a, **k)
            return wrapped
        return contextlib.nullcontext()         # with profile("phase"):


profile = _Profile()


class Profiler:
    """``Profiler(warmup_frames=, record_frames=, then=).enable()`` (code_single/tools/train.py:1437-1443, only with
    ``--profile_iters``): accepted and inert -- device timings of this path come from rocprofv3 (profiles/)."""

    def __init__(self, warmup_frames: int = 0, record_frames: int = 0, then=None):
        self.then = then

    def enable(self):
        return self

    def disable(self):
        return self
