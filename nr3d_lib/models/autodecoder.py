"""``nr3d_lib.models.autodecoder.AutoDecoderMixin`` (reference imports: app/models/asset_base.py:17,
app/models/shared/batched_neus.py:28): the latent-per-instance machinery of the shared (code_multi) models.

Restated from its call sites (the implementation lives in the absent nr3d_lib; semantics fixed here):
``autodecoder_populate(key_maps={'ins_id': [full unique ids]}, latent_maps={'z_ins': Embedding(num_objs, dim)})``
(app/models/shared/batched_neus.py:108-124, 343-380) registers the latent tables as ``self._latents[name]`` (sub-modules:
they are trained, check-pointed and moved with the model -- ``z_ins_all`` is ``self._latents['z_ins']``, :97-99) and the
key -> row maps as ``self._index_maps[key][value]`` (``set_condition`` turns instance ids into rows with them, :142-145)
and ``self._keys[key]`` (the ordered lists)."""
from typing import Dict, List

import torch.nn as nn


class AutoDecoderMixin:
    def autodecoder_populate(self, key_maps: Dict[str, List] = None, latent_maps: Dict[str, nn.Module] = None, **unused):
        assert isinstance(self, nn.Module), "AutoDecoderMixin is mixed into an nn.Module"
        key_maps = dict(key_maps or {})
        latent_maps = dict(latent_maps or {})
        lens = {len(v) for v in key_maps.values()}
        assert len(lens) <= 1, "autodecoder_populate: every key list names the same instances"
        for name, emb in latent_maps.items():
            if lens and hasattr(emb, "num_embeddings"):
                assert emb.num_embeddings == next(iter(lens)), f"latent table {name!r}: one row per instance"
        self._keys = {k: list(v) for k, v in key_maps.items()}
        self._index_maps = {k: {key: i for i, key in enumerate(v)} for k, v in key_maps.items()}
        self._latents = nn.ModuleDict(latent_maps)

    def autodecoder_latents(self, name: str):
        return self._latents[name]
