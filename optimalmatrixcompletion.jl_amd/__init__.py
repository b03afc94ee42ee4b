"""MI355X-native node-relaxation engine for OptimalMatrixCompletion.jl's hot path (HIP, gfx950)."""
from ._lib import LIB_PATH, EXPORTS, OmcError, RelaxParams, load  # noqa: F401
from .api import Engine, default_params, CUT_TYPES, DIR_CODES, BREAKPOINTS, STATUS_NAMES  # noqa: F401
from . import data, bnb, bnb_stream  # noqa: F401,E402
