"""lycoris_amd -- MI355X (gfx950) native forward/backward for the LyCORIS adapter hot path.

Scope: LoCon / LoHa / LoKr / (IA)^3 on nn.Linear and nn.Conv2d (nn.Conv1d through its Conv2d twin), forward + backward,
plus the data-parallel adapter-gradient all-reduce; nn.Conv3d in the reference's rebuild form with ATen ops (no kernel).  Everything else of LyCORIS (wrappers, presets, tools) stays upstream:
``install()`` plugs the native module classes into the reference's own registries so ``create_lycoris`` /
``lycoris.kohya.create_network`` build native adapters without any change to the caller.
"""
from __future__ import annotations

__version__ = "0.1.0"

from . import functional, modules, ops  # noqa: F401
from .modules import IA3Module, LoConModule, LohaModule, LokrModule  # noqa: F401

NATIVE_ALGOS = {"lora": LoConModule, "locon": LoConModule, "loha": LohaModule, "lokr": LokrModule, "ia3": IA3Module}
_installed = {}


def install(strict: bool = True) -> bool:
    """Rebind the reference's plug points to the native classes:

    * ``lycoris.wrapper.network_module_dict`` -- the dict object ``lycoris.kohya`` imports too (wrapper.py:45-55);
      the missing ``"ia3"`` key is added (kohya.py:105 expects it);
    * ``lycoris.modules.MODULE_LIST`` -- the from-weights registry (modules/__init__.py:19-30).

    Returns False (or raises when ``strict``) if the upstream ``lycoris`` package is not importable.  Presets and
    every other piece of upstream state are left untouched.  ``uninstall()`` restores the previous bindings.

    A variant outside the covered set (grouped convolutions, non-zero padding modes) raises
    NotImplementedError when the network is built -- nothing falls back silently, and there is no delegation to the reference's
    torch implementation.  nn.Conv3d layers are adapted by these same classes, evaluated as ``F.conv3d(x, dW)`` with ATen ops
    (modules/base.py ``_aten_only``).
    """
    try:
        import lycoris.modules as ref_modules
        import lycoris.wrapper as ref_wrapper
    except ImportError as e:
        if strict:
            raise ImportError("lycoris_amd.install(): the upstream `lycoris` package is not importable") from e
        return False
    if _installed:
        return True
    _installed["dict"] = dict(ref_wrapper.network_module_dict)
    _installed["list"] = list(ref_modules.MODULE_LIST)
    ref_wrapper.network_module_dict.update(NATIVE_ALGOS)
    by_name = {"LoConModule": LoConModule, "LohaModule": LohaModule, "LokrModule": LokrModule, "IA3Module": IA3Module}
    ref_modules.MODULE_LIST[:] = [by_name.get(cls.__name__, cls) for cls in ref_modules.MODULE_LIST]
    return True


def uninstall() -> None:
    if not _installed:
        return
    import lycoris.modules as ref_modules
    import lycoris.wrapper as ref_wrapper

    ref_wrapper.network_module_dict.clear()
    ref_wrapper.network_module_dict.update(_installed["dict"])
    ref_modules.MODULE_LIST[:] = _installed["list"]
    _installed.clear()
