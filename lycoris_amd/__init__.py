"""lycoris_amd -- MI355X (gfx950) native forward/backward for the LyCORIS adapter hot path.

Scope: LoCon / LoHa / LoKr / (IA)^3 on nn.Linear and nn.Conv2d, forward + backward, plus the
data-parallel adapter-gradient all-reduce.  Everything else of LyCORIS (wrappers, presets, tools) stays upstream:
``install()`` plugs the native module classes into the reference's own registries so ``create_lycoris`` /
``lycoris.kohya.create_network`` build native adapters without any change to the caller.
"""
from __future__ import annotations

__version__ = "0.1.0"

from . import functional, modules, ops  # noqa: F401
from .modules import IA3Module, LoConModule, LohaModule, LokrModule  # noqa: F401

NATIVE_ALGOS = {"lora": LoConModule, "locon": LoConModule, "loha": LohaModule, "lokr": LokrModule, "ia3": IA3Module}
_installed = {}


def _delegating(native_cls, ref_cls):
    """A registry entry that builds the native module and, for the variants the native path does not cover (DoRA,
    Tucker forms, Conv1d/Conv3d, ...: the constructor raises NotImplementedError), the reference's own torch module
    instead -- loudly (one warning per variant).  Class attributes / classmethods (`name`, `algo_check`,
    `make_module_from_state_dict`, ...) are the native ones."""
    import warnings

    class _Delegating(native_cls):
        _ref_cls = ref_cls
        _warned = set()

        def __new__(cls, *args, **kwargs):
            try:
                return native_cls(*args, **kwargs)  # not an instance of cls: __init__ is not run a second time
            except NotImplementedError as e:
                if cls._ref_cls is None:
                    raise
                key = str(e)
                if key not in cls._warned:
                    cls._warned.add(key)
                    warnings.warn(f"{e}  Using the reference implementation ({cls._ref_cls.__module__}."
                                  f"{cls._ref_cls.__name__}) for this layer.", RuntimeWarning, stacklevel=2)
                return cls._ref_cls(*args, **kwargs)

    _Delegating.__name__ = native_cls.__name__
    _Delegating.__qualname__ = native_cls.__qualname__
    return _Delegating


def install(strict: bool = True, delegate_unsupported: bool = False) -> bool:
    """Rebind the reference's plug points to the native classes:

    * ``lycoris.wrapper.network_module_dict`` -- the dict object ``lycoris.kohya`` imports too (wrapper.py:45-55);
      the missing ``"ia3"`` key is added (kohya.py:105 expects it);
    * ``lycoris.modules.MODULE_LIST`` -- the from-weights registry (modules/__init__.py:19-30).

    Returns False (or raises when ``strict``) if the upstream ``lycoris`` package is not importable.  Presets and
    every other piece of upstream state are left untouched.  ``uninstall()`` restores the previous bindings.

    ``delegate_unsupported``: by default a variant outside the native path (``weight_decompose``, ``use_tucker`` on k>1
    convolutions, Conv1d/Conv3d, ...) raises NotImplementedError when the network is built -- nothing falls back
    silently.  With ``delegate_unsupported=True`` such layers are built by the reference's own class (its torch
    implementation on the same device; a RuntimeWarning names the variant), so a mixed network still trains.
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
    algos = dict(NATIVE_ALGOS)
    if delegate_unsupported:
        ref_by_key = _installed["dict"]
        ref_ia3 = next((c for c in _installed["list"] if c.__name__ == "IA3Module"), None)
        algos = {k: _delegating(v, ref_by_key.get(k, ref_ia3 if k == "ia3" else None)) for k, v in algos.items()}
    ref_wrapper.network_module_dict.update(algos)
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
