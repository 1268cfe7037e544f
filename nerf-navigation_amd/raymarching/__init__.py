from .raymarching import *  # noqa: F401,F403
