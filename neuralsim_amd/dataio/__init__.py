"""Synthetic ``SceneDataset`` implementations (the reference's dataset seam, dataio/scene_dataset.py:13-74)."""
from .synthetic import SyntheticObjectDataset, SyntheticStreetDataset  # noqa: F401
