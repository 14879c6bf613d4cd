"""multimodal_mtrssm_amd -- MI355X-native MoPoE-M(MT)RSSM rollout path.

Drop-in classes for the hot path of Mamo1031/Multimodal-MTRSSM (``State`` / ``MTState``,
``Representation`` / ``Transition`` / ``MTRNN``, ``MoPoE_MRSSM`` / ``MoPoE_MMTRSSM``, ``likelihood``)
running on hand-written HIP kernels for gfx950 through the C-ABI of ``include/mtrssm.h``.
Importing the package does not need a GPU; running a rollout does, and fails loudly otherwise.
"""

from multimodal_mtrssm_amd.cnn import Decoder, Encoder
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
from multimodal_mtrssm_amd.parallel import FlatDataParallel, GlobalRowNoise
from multimodal_mtrssm_amd.state import MTState, State, cat_mtstates, cat_states, stack_mtstates, stack_states

__version__ = "0.1.0"

__all__ = [
    "MLP", "MTRNN", "Decoder", "DeviceEpisodeLoader", "Distribution", "Encoder", "EpisodeDataModule", "EpisodeDataModuleConfig", "FlatAdamW", "FlatDataParallel", "GlobalRowNoise", "MTState", "MoPoE_MMTRSSM",
    "MoPoE_MRSSM", "MultiOneHot", "MultiOneHotFactory", "ReduceLROnPlateau", "Representation", "State", "Transition", "cat_distribution",
    "cat_mtstates", "cat_states", "inject_uniforms", "kl_divergence", "likelihood", "load_reference_checkpoint", "make_mmtrssm", "make_mrssm",
    "stack_distribution", "stack_mtstates", "stack_states",
]
