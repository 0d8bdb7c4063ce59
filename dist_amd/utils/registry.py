"""Name -> class registries (reference utils/registry.py:6-65: `@X.register()`, `.get(name)`
returns None when the name is unknown, which is how MODEL.NAME 'clip' falls back to BaseVideoModel)."""


class Registry:
    def __init__(self, table_name=""):
        self._name = table_name
        self._entrys = {}

    def _register(self, name, entry):
        if name in self._entrys:
            raise KeyError(f"{name} is already registered in {self._name}")
        self._entrys[name] = entry

    def register(self, function=None, prefix=""):
        def deco(entry):
            self._register(prefix + entry.__name__, entry)
            return entry
        return deco if function is None else deco(function)

    def get(self, name):
        return self._entrys.get(name)

    def get_all_registered(self):
        return list(self._entrys)
