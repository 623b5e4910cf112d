"""``nr3d_lib.models.attributes.transform`` -- the module path code_single/tools/render.py:40 imports ``TransformMat4x4`` from;
the classes live in the package's ``__init__``."""
from . import (RotationMat3x3, RotationQuaternion, Scale, TransformMat3x4, TransformMat4x4, TransformRT,  # noqa: F401
               Translation)
