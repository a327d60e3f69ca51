"""hmc.jl_amd -- MI355X-native batched Gibbs sampler for Gaussian HMMs.

Drop-in for the data-parallel hot path of joe5saia/Hmc.jl (estimatemodel ->
gibbssample! -> gibbssweep! and the per-draw forecast), behind the reference's own
function surface (`hmc.py` mirrors module Hmc; `julia/Hmc.jl` is the Julia host
module that ccalls the same C ABI).  Compute lives in csrc/ (hand-written HIP for
gfx950) behind include/hmcg.h.  There is no CPU fallback.

Import as `import hmc_jl_amd` (root-level shim; the directory name carries a dot).
"""
from . import _lib, synth  # noqa: F401
from ._lib import HmcgError, build, load  # noqa: F401
