from .ffmlp import FFMLP, ffmlp_forward  # noqa: F401
