"""``nr3d_lib.logger.Logger`` (reference imports: app/models/asset_base.py:13, app/models/single/neus.py:22): the
tensorboard / image logger of the harness.  Out of scope here; this stand-in accepts every call the
model-side code makes (``logger.add(...)``, ``add_nested_dict``, ``add_imgs`` ...) and records scalars in memory."""


class Logger:
    def __init__(self, *args, **kwargs):
        self.scalars = {}

    def add(self, category, k, v, it=None):
        self.scalars.setdefault(f"{category}/{k}", []).append((it, float(v) if hasattr(v, "__float__") else v))

    def add_nested_dict(self, category, k=None, d=None, it=None):
        if d is None and isinstance(k, dict):
            k, d = "", k
        for kk, vv in (d or {}).items():
            if isinstance(vv, dict):
                self.add_nested_dict(category, f"{k}.{kk}" if k else kk, vv, it)
            else:
                self.add(category, f"{k}.{kk}" if k else kk, vv, it)

    def __getattr__(self, name):        # add_imgs / add_figure / add_open3d ...: accepted, not recorded
        if name.startswith("add"):
            return lambda *a, **k: None
        raise AttributeError(name)
