"""``nr3d_lib.fmt.log`` -- the console logger the reference calls as ``log.info / log.warning / log.error`` (28 files)."""
import logging
import sys

log = logging.getLogger("nr3d")
if not log.handlers:
    _h = logging.StreamHandler(sys.stdout)
    _h.setFormatter(logging.Formatter("%(asctime)s-%(levelname).4s %(message)s", datefmt="%H:%M:%S"))
    log.addHandler(_h)
    log.setLevel(logging.INFO)
    log.propagate = False


def colored_str(s, *a, **k):
    return str(s)
