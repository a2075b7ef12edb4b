"""Name -> class registries: the plug-in boundary the model constructor looks classes up in
(reference: pipeline/registry.py:9-55).  Same dict names and keys; the values are the
HIP-backed classes of this package, so an existing YAML is a drop-in."""
import torch

from distributions.noise import SourceUniform
from model.bridges import UniformFlow, SchrodingerBridge, TelegraphBridge
from model.thermostats import (ConstantThermostat, InverseThermostat, LinearThermostat, InverseSquareThermostat,
                               SigmoidThermostat)
from encoders.multimodal_epic import UniModalEPiC, MultiModalEPiC
from encoders.particle_transformer import MultiModalParticleTransformer
from encoders.mlp import MultiModalMLP
from distributions.toys import NGaussians, TwoMoons

from datamodules.aoj import AspenOpenJets

registered_datasets = {"AspenOpenJets": AspenOpenJets}     # datamodules/aoj.py

registered_distributions = {"SourceUniform": SourceUniform,
                            "NGaussians": NGaussians, "TwoMoons": TwoMoons}     # toys.py (not registered upstream: torchdyn)

registered_bridges = {"UniformFlow": UniformFlow, "SchrodingerBridge": SchrodingerBridge,
                      "TelegraphBridge": TelegraphBridge}

registered_models = {"UniModalEPiC": UniModalEPiC, "MultiModalEPiC": MultiModalEPiC,
                     "MultiModalParticleTransformer": MultiModalParticleTransformer,
                     "MultiModalMLP": MultiModalMLP}      # the toy notebook's class (data_toys.ipynb cell 3)

registered_thermostats = {"ConstantThermostat": ConstantThermostat, "LinearThermostat": LinearThermostat,
                          "InverseThermostat": InverseThermostat,
                          "InverseSquareThermostat": InverseSquareThermostat,
                          "SigmoidThermostat": SigmoidThermostat}

registered_optimizers = {"Adam": torch.optim.Adam, "AdamW": torch.optim.AdamW}

registered_schedulers = {"StepLR": torch.optim.lr_scheduler.StepLR,
                         "ReduceLROnPlateau": torch.optim.lr_scheduler.ReduceLROnPlateau,
                         "CosineAnnealingLR": torch.optim.lr_scheduler.CosineAnnealingLR,
                         "OneCycleLR": torch.optim.lr_scheduler.OneCycleLR}
