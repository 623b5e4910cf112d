from .sky import SimpleSky  # noqa: F401
