"""Architecture discovery with the reference's contract
(/root/reference/basicsr/models/archs/__init__.py:9-46): every ``*_arch.py`` file in this
folder is imported and ``define_network(opt)`` pops ``type`` and instantiates the first
module attribute of that name with the remaining keys; unknown types raise ValueError.

The arch classes are also registered with ``@ARCH_REGISTRY.register()`` (``refid_amd.registry``: the surface of later
BasicSR releases, which the reference's vintage does not have); ``define_network`` asks the registry first."""
import importlib
import os

from ..registry import ARCH_REGISTRY

_arch_folder = os.path.dirname(os.path.abspath(__file__))
_arch_filenames = sorted(os.path.splitext(f)[0] for f in os.listdir(_arch_folder) if f.endswith('_arch.py'))
_arch_modules = [importlib.import_module(f'{__name__}.{n}') for n in _arch_filenames]


def dynamic_instantiation(modules, cls_type, opt):
    cls_ = None
    for module in modules:
        cls_ = getattr(module, cls_type, None)
        if cls_ is not None:
            break
    if cls_ is None:
        raise ValueError(f'{cls_type} is not found.')
    return cls_(**opt)


def define_network(opt):
    network_type = opt.pop('type')
    if network_type in ARCH_REGISTRY:
        return ARCH_REGISTRY.get(network_type)(**opt)
    return dynamic_instantiation(_arch_modules, network_type, opt)
