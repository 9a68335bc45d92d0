"""lightweaver_amd -- MI355X (gfx950) implementation of Lightweaver's formal_sol_gamma_matrices
hot path behind the C ABI of include/lwhip.h.  See DESIGN.md."""
from . import _abi as abi  # noqa: F401
from .model import AtomData, Boundary, Problem, TransitionData  # noqa: F401

__all__ = ['abi', 'AtomData', 'Boundary', 'Problem', 'TransitionData']
