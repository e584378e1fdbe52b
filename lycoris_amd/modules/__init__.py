"""Native adapter modules (same class names and registry protocol as lycoris/modules/__init__.py:19-46)."""
import torch

from .base import LycorisBaseModule
from .ia3 import IA3Module
from .locon import LoConModule
from .loha import LohaModule
from .lokr import LokrModule

MODULE_LIST = [LoConModule, LohaModule, IA3Module, LokrModule]


def get_module(lyco_state_dict, lora_name):
    for module in MODULE_LIST:
        if module.algo_check(lyco_state_dict, lora_name):
            return module, tuple(module.extract_state_dict(lyco_state_dict, lora_name))
    return None, None


@torch.no_grad()
def make_module(lyco_type, params, lora_name, orig_module):
    if isinstance(orig_module, torch.nn.Conv1d):  # checkpoint tensors [.., .., k] -> the native [.., .., 1, k] (modules/base.py _Conv1dTwin)
        from .base import _lift1d
        params = tuple(_lift1d(p) for p in params)
    try:
        return lyco_type.make_module_from_state_dict(lora_name, orig_module, *params)
    except NotImplementedError:
        return None
