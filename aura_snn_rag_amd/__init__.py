"""aura_snn_rag_amd -- MI355X-native (gfx950) implementation of Aura's SNN-timestep +
episodic-retrieval hot path, behind the reference's own module API.

    from aura_snn_rag_amd.core.hippocampal import HippocampalFormation
    from aura_snn_rag_amd.base.neuron import IzhikevichNeuron, AdExNeuron, VectorizedLIFNeuron
    from aura_snn_rag_amd.core.language_zone.gif_neuron import GIFNeuron
    from aura_snn_rag_amd.core.language_zone.snn_ffn import SNNFFN, HybridFFN

All arithmetic on the path runs in ``lib/libaura_hip.so`` (hand-written HIP, C ABI in
``include/aura_hip.h``); there is no CPU or eager-PyTorch fallback.
"""
from ._lib import AuraHipError, AuraHipUnavailable, load as load_library  # noqa: F401

__version__ = "0.1.0"
