"""em_model_manned_bayes_amd -- MI355X-native sampling path of the MIT-LL Bayesian-network airspace
encounter models (reference: Airspace-Encounter-Models/em-model-manned-bayes), behind the
reference's own function and class names.  The compute path is libemgpu.so (HIP, gfx950);
importing a sampling entry point without it raises ImportError, and creating a Context without a
GPU raises EmgpuError(EMGPU_ERR_NO_DEVICE): there is no CPU fallback.

(The directory is spelled with underscores because Python cannot import a hyphenated name.)
"""
from ._lib import EmgpuError  # noqa: F401
from .em_io import em_read, em_write, Parms  # noqa: F401
from .functions import (rng, asub2ind, aind2sub, bn_sort, bn_sample, dbn_sample, dbn_hierarchical_sample,  # noqa: F401
                        bn_dirichlet_prior, setTransitionPriors, discretize_bayes, hierarchical_cutpoints,
                        hierarchical_discretize, events2samples, events2controls)
from .encounter_model import EncounterModel, EncounterModelEvents, UncorEncounterModel, CorTerminalModel  # noqa: F401
from .native import Context, NativeModel, default_context  # noqa: F401
from .legacy import em_sample, sample2track  # noqa: F401

__all__ = ["em_read", "em_write", "rng", "asub2ind", "aind2sub", "bn_sort", "bn_sample", "dbn_sample",
           "dbn_hierarchical_sample", "bn_dirichlet_prior", "setTransitionPriors", "discretize_bayes",
           "hierarchical_cutpoints", "hierarchical_discretize", "events2samples", "events2controls",
           "EncounterModel", "EncounterModelEvents", "UncorEncounterModel", "CorTerminalModel",
           "em_sample", "sample2track", "Context", "NativeModel", "EmgpuError"]
