"""ExperimentConfigs: YAML -> nested dataclasses, the object handed to every hot-path
constructor.  Field names, defaults and derived dimensions follow the reference
(pipeline/configs.py:17-148) so its YAML files load unchanged.  The Comet section is
accepted (and ignored) for file compatibility; there is no logger service here and no
credential default.
"""
import os
from dataclasses import dataclass, field, asdict, fields
from typing import Dict, List, Optional, Union

import yaml


@dataclass
class CometLoggerConfig:
    project_name: str = None
    save_dir: Optional[str] = None
    experiment_name: str = None
    workspace: str = None
    api_key: str = None


@dataclass
class DataConfig:
    modality: str = None
    batch_size: int = None
    split_ratios: List[float] = None
    transform: str = None
    continuous_features: List[str] = None
    discrete_features: str = None
    target_name: str = None
    target_path: str = None
    target_train_files: Optional[List[str]] = None
    target_test_files: Optional[List[str]] = None
    source_name: str = None
    source_path: str = None
    source_train_files: Optional[List[str]] = None
    source_test_files: Optional[List[str]] = None
    num_jets: Optional[int] = None
    min_num_particles: int = None
    max_num_particles: int = None
    dim_continuous: Optional[int] = 0
    dim_discrete: Optional[int] = 0
    dim_context_continuous: Optional[int] = 0
    dim_context_discrete: Optional[int] = 0
    vocab_size: Optional[int] = 0
    vocab_freq: Optional[List[float]] = None
    vocab_size_context: Optional[int] = 0
    num_workers: Optional[int] = 0
    pin_memory: Optional[bool] = False


@dataclass
class EncoderConfig:
    name: str
    num_blocks: Optional[Union[int, List[int]]]
    dim_hidden_local: Optional[Union[int, List[int]]]
    num_heads: Optional[int] = None
    dim_hidden_glob: Optional[Union[int, List[int]]] = None
    skip_connection: Optional[bool] = True
    dropout: Optional[float] = 0.0
    data_augmentation: Optional[bool] = False
    dim_emb_time: Optional[int] = 0
    dim_emb_continuous: Optional[int] = 0
    dim_emb_discrete: Optional[int] = 0
    dim_emb_context_continuous: Optional[int] = 0
    dim_emb_context_discrete: Optional[int] = 0


@dataclass
class TrainerConfig:
    max_epochs: int
    optimizer_name: str
    optimizer_params: Dict[str, Union[float, List[float], bool]]
    scheduler_name: Optional[str] = None
    scheduler_params: Optional[Dict[str, Union[float, List[float], bool]]] = None
    gradient_clip_val: float = 1.0


@dataclass
class ModelConfig:
    bridge_continuous: Optional[str] = None
    bridge_discrete: Optional[str] = None
    solver_continuous: Optional[str] = None
    solver_discrete: Optional[str] = None
    sigma: Optional[float] = None
    gamma: Optional[float] = None
    thermostat_fn: Optional[str] = None
    loss_weights: Optional[Union[str, List[float]]] = "fixed"
    num_timesteps: int = None
    time_eps: float = 0.0
    topk: int = None
    save_path_history_snapshots: Optional[List[int]] = None


@dataclass
class CheckpointsConfig:
    dirpath: Optional[str] = None
    monitor: str = "val_loss"
    mode: str = "min"
    save_top_k: int = 1
    filename: str = "best"
    save_last: bool = True
    patience: Optional[int] = None
    stopping_threshold: Optional[float] = None


_SECTIONS = {"data": DataConfig, "encoder": EncoderConfig, "trainer": TrainerConfig, "model": ModelConfig,
             "checkpoints": CheckpointsConfig, "comet_logger": CometLoggerConfig}


class ExperimentConfigs:
    """ExperimentConfigs(path_to_yaml | dict).  Sections: data, encoder, trainer, model,
    checkpoints, comet_logger (the last two optional here)."""

    def __init__(self, config):
        self.path = None
        if isinstance(config, str):
            with open(config, "r") as fh:
                config = yaml.safe_load(fh)
        for name, cls in _SECTIONS.items():
            section = config.get(name)
            if section is None:
                if name in ("checkpoints", "comet_logger"):
                    section = {}
                else:
                    raise KeyError(f"config section '{name}' is required")
            setattr(self, name, cls(**section))           # unknown keys raise TypeError, as upstream
        assert self.encoder.dim_emb_time > 0, "Non-zero dim_emb_time must be provided."

        d = self.data
        if d.continuous_features is not None:
            d.dim_continuous = len(d.continuous_features)
        if d.discrete_features is not None:
            assert d.vocab_size > 0, "vocab_size must be provided for discrete features"
        if d.discrete_features == "onehot":
            d.dim_continuous += d.vocab_size
        if d.modality in ("multi-modal", "discrete"):
            d.dim_discrete = 1

    def update(self, config):
        if not isinstance(config, dict):
            return
        for key, val in config.items():
            if key == "path":
                self.path = val
            else:
                section = getattr(self, key)
                for sub_key, sub_val in val.items():
                    setattr(section, sub_key, sub_val)

    def to_dict(self):
        out = {"path": self.path}
        for name in _SECTIONS:
            out[name] = asdict(getattr(self, name))
        return out

    def save(self, path, name="config.yaml"):
        target = os.path.join(path, name)
        if not os.path.exists(target):
            with open(target, "w") as fh:
                yaml.dump(self.to_dict(), fh, default_flow_style=False)

    def print(self, indent=0):
        for sec, vals in self.to_dict().items():
            print(" " * indent + f"{sec}:")
            if isinstance(vals, dict):
                for k, v in vals.items():
                    print(" " * (indent + 4) + f"{k}: {v}")
            else:
                print(" " * (indent + 4) + f"{vals}")
