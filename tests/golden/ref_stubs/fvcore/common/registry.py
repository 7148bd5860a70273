"""Stub of fvcore.common.registry for importing the reference offline (make_golden.py only)."""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._obj_map[o.__name__] = o
                return o
            return deco
        self._obj_map[obj.__name__] = obj
        return obj

    def get(self, name):
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map
