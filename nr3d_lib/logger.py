"""``nr3d_lib.logger.Logger`` (reference imports: code_single/tools/train.py:37, 1218-1224; app/models/asset_base.py:13):
the tensorboard / image logger of the harness.  This stand-in keeps scalars in memory (``save_stats`` / ``load_stats``
pickle them next to the experiment, as the trainer expects) and accepts every image / figure call without writing."""
import os
import pickle


class Logger:
    def __init__(self, root: str = None, img_root: str = None, monitoring: str = None, monitoring_dir: str = None,
                 rank: int = 0, is_master: bool = True, multi_process_logging: bool = False, **unused):
        self.root, self.rank, self.is_master = root, rank, is_master
        self.scalars = {}
        self.stats = {}

    def add(self, category, k, v, it=None):
        try:
            v = float(v)
        except (TypeError, ValueError):
            pass
        self.scalars.setdefault(f"{category}/{k}", []).append((it, v))

    def add_nested_dict(self, category, k=None, d=None, it=None):
        if d is None and isinstance(k, dict):
            k, d = "", k
        for kk, vv in (d or {}).items():
            if isinstance(vv, dict):
                self.add_nested_dict(category, f"{k}.{kk}" if k else kk, vv, it)
            else:
                self.add(category, f"{k}.{kk}" if k else kk, vv, it)

    def save_stats(self, filename: str = "stats.p"):
        if self.root and self.is_master:
            with open(os.path.join(self.root, filename), "wb") as f:
                pickle.dump(self.scalars, f)

    def load_stats(self, filename: str = "stats.p"):
        path = os.path.join(self.root, filename) if self.root else None
        if path and os.path.exists(path):
            with open(path, "rb") as f:
                self.scalars = pickle.load(f)

    def __getattr__(self, name):        # add_imgs / add_figure / add_open3d / add_text ...: accepted, not recorded
        if name.startswith("add") or name in ("close", "flush"):
            return lambda *a, **k: None
        raise AttributeError(name)


from .fmt import log  # noqa: E402,F401
