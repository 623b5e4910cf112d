"""``nr3d_lib.config.ConfigDict`` -- the attribute-access dict the renderers build their per-query configs from
(``ConfigDict(**model.ray_query_cfg, **config)``, app/renderers/single_volume_renderer.py:241) -- and
``parse_device_ids``.  The YAML front end of nr3d_lib.config (``${...}`` / ``${eval:...}`` resolution, CLI merging) is
harness, out of scope."""
from typing import List, Union

import torch


class ConfigDict(dict):
    """dict with attribute access; nested plain dicts are wrapped on construction / assignment."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return ConfigDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(ConfigDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k) from None

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]

    def copy(self):
        return ConfigDict(self)

    def deepcopy(self):
        import copy
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        import copy
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self):
        def un(v):
            if isinstance(v, ConfigDict):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(un(x) for x in v)
            return v
        return un(self)


def parse_device_ids(device_ids: Union[str, int, List[int]] = -1, to_torch: bool = False):
    """-1 / 'all' -> every visible device; an int, a list or a comma-separated string -> that list."""
    if isinstance(device_ids, str):
        device_ids = -1 if device_ids in ("all", "-1") else [int(x) for x in device_ids.split(",") if x != ""]
    if isinstance(device_ids, int):
        device_ids = list(range(torch.cuda.device_count())) if device_ids == -1 else [device_ids]
    device_ids = list(device_ids)
    return [torch.device("cuda", i) for i in device_ids] if to_torch else device_ids


# ------------------------------------------------------------------------------------------------ YAML front end
def _lookup(root: dict, path: str):
    cur = root
    for k in path.split("."):
        cur = cur[k]
    return cur


def _safe_arith(expr: str):
    """``${eval:...}`` of the reference's configs is arithmetic (``2**20``, ``8*(2**20)``, ``500+1000``, ``int(1.5*4096)``):
    evaluated by walking the AST -- numbers, + - * / // % **, unary minus, comparisons-free, and calls of
    min / max / int / float / round / abs only.  (Python's ``eval`` with empty builtins is not a sandbox; a YAML file from
    somewhere else must not be able to run code here.)"""
    import ast
    import operator as op
    bin_ops = {ast.Add: op.add, ast.Sub: op.sub, ast.Mult: op.mul, ast.Div: op.truediv, ast.FloorDiv: op.floordiv,
               ast.Mod: op.mod, ast.Pow: op.pow}
    un_ops = {ast.USub: op.neg, ast.UAdd: op.pos}
    funcs = {"min": min, "max": max, "int": int, "float": float, "round": round, "abs": abs}

    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float, bool)):
            return n.value
        if isinstance(n, ast.BinOp) and type(n.op) in bin_ops:
            a, b = ev(n.left), ev(n.right)
            if isinstance(n.op, ast.Pow):
                # bound the SIZE of the result, not only the exponent: ((2**4096)**4096)**4096 would allocate gigabytes
                if abs(b) > 4096 or (isinstance(a, int) and isinstance(b, int) and abs(a).bit_length() * abs(b) > 4096):
                    raise ValueError("${eval:...}: power too large")
            r = bin_ops[type(n.op)](a, b)
            if isinstance(r, int) and abs(r).bit_length() > 8192:
                raise ValueError("${eval:...}: integer too large")
            return r
        if isinstance(n, ast.UnaryOp) and type(n.op) in un_ops:
            return un_ops[type(n.op)](ev(n.operand))
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id in funcs and not n.keywords:
            return funcs[n.func.id](*[ev(a) for a in n.args])
        if isinstance(n, (ast.Tuple, ast.List)):
            return [ev(e) for e in n.elts]
        raise ValueError(f"${{eval:...}}: unsupported expression element {type(n).__name__} in {expr!r}")
    return ev(ast.parse(str(expr).strip(), mode="eval"))


def _resolve_str(s: str, root: dict, depth: int = 0, here=()):
    """``${a.b}`` -> the value at that path of the ROOT config (typed when the string is nothing but the reference),
    ``${eval:"expr"}`` / ``${eval:expr}`` -> Python arithmetic -- the two interpolation forms the reference's configs
    use on the model blocks (lotd_neus.dtu.230814.yaml:82-166)."""
    import re
    assert depth < 16, f"circular interpolation in {s!r}"
    pat = re.compile(r"\$\{([^{}]+)\}")
    m = pat.fullmatch(s.strip())

    def value(body):
        if body.startswith("eval:"):
            expr = body[5:].strip()
            if len(expr) >= 2 and expr[0] == expr[-1] and expr[0] in "\"'":
                expr = expr[1:-1]
            expr = _resolve_str(expr, root, depth + 1, here) if "${" in expr else expr
            return _safe_arith(expr)
        where = here
        if body.startswith("."):        # relative: ``${.sibling}`` / ``${..uncle}`` (``Cyclist: ${.Pedestrian}``, all_occ.240201.yaml:599)
            n = len(body) - len(body.lstrip("."))
            base = list(here[:len(here) - n]) if n <= len(here) else []
            rest = body.lstrip(".")
            where = tuple(base + rest.split("."))
            body = ".".join(str(k) for k in where)
        v = _lookup(root, body)
        return _resolve_str(v, root, depth + 1, tuple(where) if where else here) if isinstance(v, str) and "${" in v else v
    if m:
        return value(m.group(1))
    return pat.sub(lambda mm: str(value(mm.group(1))), s)


def resolve_config(cfg: dict) -> ConfigDict:
    """Resolve every interpolation of a loaded YAML tree against its own root."""
    def walk(v, here=()):
        if isinstance(v, dict):
            return {k: walk(x, here + (k,)) for k, x in v.items()}
        if isinstance(v, list):
            return [walk(x, here + (i,)) for i, x in enumerate(v)]
        if isinstance(v, str) and "${" in v:
            try:
                return walk(_resolve_str(v, cfg, 0, here), here)
            except (KeyError, TypeError):
                return v            # references into parts of the harness config that are not present stay verbatim
        return v
    return ConfigDict(walk(cfg))


def load_config(path: str) -> ConfigDict:
    import yaml
    with open(path) as f:
        return resolve_config(yaml.safe_load(f))


def save_config(cfg, path: str):
    import yaml
    with open(path, "w") as f:
        yaml.safe_dump(cfg.to_dict() if isinstance(cfg, ConfigDict) else dict(cfg), f, sort_keys=False)


def _set_by_path(root: dict, path: str, value):
    cur = root
    keys = path.split(".")
    for k in keys[:-1]:
        if k not in cur or not isinstance(cur[k], dict):
            cur[k] = {}
        cur = cur[k]
    cur[keys[-1]] = value


class BaseConfig:
    """The command-line front end of the reference's tools (code_single/tools/train.py:1691-1698: ``bc = BaseConfig();
    bc.parser.add_argument(...); args = bc.parse()``): ``--config file.yaml`` (or ``--resume_dir exp_dir``), registered
    extra options, and free-form dotted overrides ``--a.b.c=value`` / ``--a.b.c value`` (YAML-typed) merged into the
    tree BEFORE the ``${...}`` interpolations are resolved; ``exp_dir`` defaults to ``./logs/<config name>``; ``ddp`` /
    ``device_ids`` are filled in as the trainer expects."""

    def __init__(self):
        import argparse
        self.parser = argparse.ArgumentParser()
        self.parser.add_argument("--config", type=str, default=None)
        self.parser.add_argument("--resume_dir", type=str, default=None)
        self.parser.add_argument("--exp_dir", type=str, default=None)
        self.parser.add_argument("--device_ids", type=str, default=None)
        self.parser.add_argument("--ddp", action="store_true")
        self.parser.add_argument("--port", type=int, default=None)

    def parse(self, argv=None, print_config: bool = True) -> ConfigDict:
        import os
        import yaml
        known, extra = self.parser.parse_known_args(argv)
        path = known.config
        if path is None and known.resume_dir is not None:
            path = os.path.join(known.resume_dir, "config.yaml")
        assert path is not None, "--config <yaml> (or --resume_dir <exp_dir>) is required"
        with open(path) as f:
            tree = yaml.safe_load(f)
        i = 0
        while i < len(extra):
            tok = extra[i]
            assert tok.startswith("--"), f"unrecognised argument {tok!r}"
            if "=" in tok:
                k, v = tok[2:].split("=", 1)
                i += 1
            else:
                k, v = tok[2:], extra[i + 1]
                i += 2
            _set_by_path(tree, k, yaml.safe_load(v))
        builtin = ("config", "resume_dir", "exp_dir", "device_ids", "ddp", "port")
        for k, v in vars(known).items():
            if k in ("config", "resume_dir") or (k == "ddp" and not v and "ddp" in tree):
                continue
            if v is None:
                # an option a TOOL registered (``bc.parser.add_argument("--cam_id", default=None)``, code_single/tools/eval.py:
                # 606-634) exists on the parsed config with its default, None included -- eval.py:143 reads ``args.cam_id``
                if k not in builtin:
                    tree.setdefault(k, None)
                continue
            tree[k] = v
        tree.setdefault("ddp", False)
        if isinstance(tree.get("device_ids"), str) or tree.get("device_ids") is None:
            tree["device_ids"] = parse_device_ids(tree.get("device_ids", -1) if tree.get("device_ids") is not None else -1) or [0]
        elif isinstance(tree["device_ids"], int):
            tree["device_ids"] = parse_device_ids(tree["device_ids"]) or [0]
        if tree.get("exp_dir") is None:
            tree["exp_dir"] = os.path.join("./logs", os.path.splitext(os.path.basename(path))[0])
        cfg = resolve_config(tree)
        if print_config:
            print(yaml.safe_dump(cfg.to_dict(), sort_keys=False))
        return cfg
