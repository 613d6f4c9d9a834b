"""Neuromorphic brain zone: AdditionLinear in-projection -> neuron groups -> AdditionLinear out.

Drop-in for the tensor contract of ``src/base/snn_brain_zones.py`` (``SpikingNeuronConfig``,
``BrainZoneConfig``, ``EnhancedSpikingNeuron``, ``NeuromorphicBrainZone``): the zone splits the
projected input into groups, runs each group through its Izhikevich / AdEx / LIF population and
concatenates the spikes.  Every arithmetic step is a HIP kernel (``aura_addition_linear``,
``aura_izh_run_*``, ``aura_adex_run_*``, ``aura_lif_run``).  The reference's event bus is accepted
and ignored on the hot path (its broadcast is host-side bookkeeping).
"""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from ..maths.addition_linear import AdditionLinear
from .neuron import AdExNeuron, IzhikevichNeuron, VectorizedLIFNeuron


class BrainZoneType(Enum):
    PREFRONTAL_CORTEX = "prefrontal_cortex"; TEMPORAL_CORTEX = "temporal_cortex"
    HIPPOCAMPUS = "hippocampus"; CEREBELLUM = "cerebellum"; THALAMUS = "thalamus"
    AMYGDALA = "amygdala"; BASAL_GANGLIA = "basal_ganglia"; BRAINSTEM = "brainstem"
    OCCIPITAL_CORTEX = "occipital_cortex"; PARIETAL_CORTEX = "parietal_cortex"
    INSULAR_CORTEX = "insular_cortex"


@dataclass
class SpikingNeuronConfig:
    neuron_type: str
    structure: str
    neurotransmitter: str
    percentage: float
    threshold: float = 0.6
    membrane_time_constant: float = 10.0
    init_surrogate_slope: float = 15.0
    beta_decay: float = 0.95
    a: float = None
    b: float = None
    c: float = None
    d: float = None
    dt: float = 0.2
    model_type: str = None
    model_params: Dict = None


@dataclass
class BrainZoneConfig:
    name: str = ""
    max_neurons: int = 1024
    min_neurons: int = 256
    neuron_type: str = "liquid"
    gated: bool = False
    num_layers: int = 2
    base_layer_container_config: Any = None
    zone_type: BrainZoneType = None
    d_model: int = 1024
    use_spiking: bool = True
    spiking_configs: List[SpikingNeuronConfig] = None
    event_bus: Any = None


class EnhancedSpikingNeuron(nn.Module):
    """One neuron population; dispatches on the config exactly as the reference
    (``snn_brain_zones.py:38-59``): ``a`` set -> Izhikevich, ``model_type == 'adex'`` -> AdEx,
    otherwise vectorised LIF."""

    def __init__(self, config, d_model, event_bus=None, zone_name=None):
        super().__init__()
        self.config = config
        self.d_model = d_model
        if config.a is not None:
            self.core = IzhikevichNeuron(a=config.a, b=config.b, c=config.c, d=config.d, dt=config.dt)
            self.mode = 'izh'
        elif config.model_type == 'adex':
            self.core = AdExNeuron(**(config.model_params or {}))
            self.mode = 'adex'
        else:
            self.core = VectorizedLIFNeuron(size=d_model, beta=config.beta_decay,
                                            threshold=config.threshold,
                                            init_slope=config.init_surrogate_slope,
                                            event_bus=event_bus, name=zone_name)
            self.mode = 'lif'
        self.register_buffer('homeo_i', torch.tensor(0.0))

    def forward(self, x):
        is_seq = x.dim() == 3
        x_eff = x + self.homeo_i
        if self.mode in ('izh', 'adex'):
            if is_seq:
                return self.core.forward_sequence(x_eff), {}, {}
            # [B, D] -> one timestep per neuron: [B, 1, D] sequence (ref :69-71)
            return self.core.forward_sequence(x_eff.unsqueeze(1)).squeeze(1), {}, {}
        # LIF.  When the reference would build a graph (grad mode on and the current or the surrogate's slope
        # requires grad) the steps go through the recording LIF step (aura_lif_train_forward / aura_lif_backward), one
        # call per timestep as the reference's loop (:73-79); otherwise the whole sequence is one launch.
        record = torch.is_grad_enabled() and (x_eff.requires_grad or self.core.slope.requires_grad)
        if is_seq:
            if not record:
                return self.core.forward_sequence(x_eff), None, {}
            spikes = [self.core(x_eff[:, t])[0] for t in range(x_eff.shape[1])]
            return torch.stack(spikes, dim=1), None, {}
        spikes, mem = self.core(x_eff)
        return spikes, mem, {}


class NeuromorphicBrainZone(nn.Module):
    def __init__(self, config: BrainZoneConfig):
        super().__init__()
        self.config = config
        self.neuron_groups = nn.ModuleDict()
        self.neuron_counts = {}
        total = max(1, config.max_neurons)
        remaining = total
        configs = config.spiking_configs or [SpikingNeuronConfig(
            neuron_type="pyramidal_default", structure="standard", neurotransmitter="glutamate",
            percentage=100.0, threshold=0.5)]
        for i, cfg in enumerate(configs):
            count = remaining if i == len(configs) - 1 else max(1, int(total * cfg.percentage / 100.0))
            count = min(count, remaining)
            if count <= 0:
                continue
            remaining -= count
            self.neuron_counts[cfg.neuron_type] = count
            self.neuron_groups[cfg.neuron_type] = EnhancedSpikingNeuron(cfg, count, config.event_bus,
                                                                         config.name)
        if len(self.neuron_groups) == 0:
            self.neuron_counts["fallback"] = total
            self.neuron_groups["fallback"] = EnhancedSpikingNeuron(configs[0], total, config.event_bus,
                                                                    config.name)
        self.input_projection = AdditionLinear(config.d_model, total, bias=False)
        self.output_projection = AdditionLinear(total, config.d_model, bias=False)

    def forward(self, x: torch.Tensor, context: Optional[Dict] = None):
        zone_input = self.input_projection(x)
        outputs, start = [], 0
        for name, module in self.neuron_groups.items():
            count = self.neuron_counts[name]
            if count <= 0:
                continue
            group_input = zone_input[..., start:start + count].contiguous()
            spikes, _, _ = module(group_input)
            outputs.append(spikes)
            start += count
        if not outputs:
            return torch.zeros_like(x), {'zone_name': self.config.name, 'error': 'no_output'}
        combined = torch.cat(outputs, dim=-1)
        output = self.output_projection(combined)
        with torch.no_grad():
            avg_rate = combined.float().mean().item()
        return output, {'zone_name': self.config.name, 'avg_firing_rate': avg_rate}
