"""stan4bart_amd — MI355X-native implementation of stan4bart's blocked Gibbs hot path.

The product is the HIP library behind ``include/stan4bart_amd.h`` (``stan4bart_amd/csrc``);
this package is the host-side mirror of the reference's R fit driver for that path.
"""
from .abi import Sampler, SamplerArgs  # noqa: F401
from .fit import GroupTerm, make_sampler_args, stan4bart_fit, fit_worker  # noqa: F401
from .friedman import generate_friedman_data  # noqa: F401
from .generics import Stan4bartFit, stan4bart  # noqa: F401
from .rcompat import RRng  # noqa: F401
