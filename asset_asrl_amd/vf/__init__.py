"""``asset_asrl_amd.vf`` -- the VectorFunctions surface needed to define ODEs (see functions.py)."""
from .functions import *  # noqa: F401,F403
from .functions import VectorFunction, MatrixFunction, Arguments  # noqa: F401
