"""``ARCH_REGISTRY`` / ``@ARCH_REGISTRY.register()`` surface (BASELINE.json north_star).

The reference's BasicSR vintage (1.2.0) has NO registry: it scans ``basicsr/models/archs/*_arch.py`` and resolves
``network_g.type`` by ``getattr`` (/root/reference/basicsr/models/archs/__init__.py:9-46).  Later
BasicSR releases replaced the scan by ``ARCH_REGISTRY`` with a ``register()`` decorator and ``get(name)``.  Both
contracts are offered: ``refid_amd.archs.define_network`` asks this registry first and falls back to the module scan,
so a maintainer of either vintage finds the surface they bind to."""


class Registry:
    """name -> class map with BasicSR's later interface: ``@REG.register()`` (or bare ``@REG.register``),
    ``REG.get(name)`` (KeyError when missing), ``name in REG``, ``REG.keys()``; a second registration of one name raises."""

    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        if name in self._obj_map:
            raise AssertionError(f"An object named '{name}' was already registered in '{self._name}' registry!")
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:                                  # @REG.register()
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)             # @REG.register  /  REG.register(cls)
        return obj

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def keys(self):
        return self._obj_map.keys()


ARCH_REGISTRY = Registry("arch")
MODEL_REGISTRY = Registry("model")
