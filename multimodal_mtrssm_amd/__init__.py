"""multimodal_mtrssm_amd -- MI355X-native MoPoE-M(MT)RSSM rollout path.

Drop-in classes for the hot path of Mamo1031/Multimodal-MTRSSM (``State`` / ``MTState``,
``Representation`` / ``Transition`` / ``MTRNN``, ``MoPoE_MRSSM`` / ``MoPoE_MMTRSSM``, ``likelihood``)
running on hand-written HIP kernels for gfx950 through the C-ABI of ``include/mtrssm.h``.
Importing the package does not need a GPU; running a rollout does, and fails loudly otherwise.
"""

import os as _os

# Effective only if HIP has not been initialised yet (harmless otherwise): with core.BRANCH_STREAMS the two modality
# branches run on two streams (core.fork_join); with ROCm's default of 4 hardware queues per process a side stream can
# share the default stream's queue once RCCL has taken its own, and the branches then serialise.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from multimodal_mtrssm_amd.cnn import Decoder, Encoder  # noqa: E402
from multimodal_mtrssm_amd.core import MoPoE_MMTRSSM, MoPoE_MRSSM
from multimodal_mtrssm_amd.dataset import DeviceEpisodeLoader, EpisodeDataModule, EpisodeDataModuleConfig
from multimodal_mtrssm_amd.distributions import (
    Distribution,
    MultiOneHot,
    MultiOneHotFactory,
    cat_distribution,
    inject_uniforms,
    kl_divergence,
    stack_distribution,
)
from multimodal_mtrssm_amd.factory import make_mmtrssm, make_mrssm
from multimodal_mtrssm_amd.networks import MLP, MTRNN, Representation, Transition
from multimodal_mtrssm_amd.objective import likelihood
from multimodal_mtrssm_amd.optim import FlatAdamW, ReduceLROnPlateau, load_reference_checkpoint
from multimodal_mtrssm_amd.parallel import FlatDataParallel
from multimodal_mtrssm_amd.state import MTState, State, cat_mtstates, cat_states, stack_mtstates, stack_states

__version__ = "0.1.0"

__all__ = [
    "MLP", "MTRNN", "Decoder", "DeviceEpisodeLoader", "Distribution", "Encoder", "EpisodeDataModule", "EpisodeDataModuleConfig", "FlatAdamW", "FlatDataParallel", "MTState", "MoPoE_MMTRSSM",
    "MoPoE_MRSSM", "MultiOneHot", "MultiOneHotFactory", "ReduceLROnPlateau", "Representation", "State", "Transition", "cat_distribution",
    "cat_mtstates", "cat_states", "inject_uniforms", "kl_divergence", "likelihood", "load_reference_checkpoint", "make_mmtrssm", "make_mrssm",
    "stack_distribution", "stack_mtstates", "stack_states",
]
