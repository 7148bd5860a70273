"""Name -> class registry with the surface the reference uses from
`fvcore.common.registry.Registry` (modules/build.py:6-31, model/build.py:5-18):
`@REG.register()` decorator (or `REG.register(obj)`), `REG.get(name)`, `name in REG`, `REG._name`.
fvcore is not installable here, and the drop-in must not depend on it."""
from __future__ import annotations

from typing import Any, Callable, Dict, Iterator, Optional, Tuple


class Registry:
    def __init__(self, name: str) -> None:
        self._name = name
        self._obj_map: Dict[str, Any] = {}

    def _do_register(self, name: str, obj: Any) -> None:
        if name in self._obj_map:
            raise AssertionError(f"An object named '{name}' was already registered in '{self._name}' registry!")
        self._obj_map[name] = obj

    def register(self, obj: Any = None) -> Any:
        if obj is None:
            def deco(func_or_class: Any) -> Any:
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name: str) -> Any:
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name: str) -> bool:
        return name in self._obj_map

    def __iter__(self) -> Iterator[Tuple[str, Any]]:
        return iter(self._obj_map.items())

    def __repr__(self) -> str:
        return f"Registry of {self._name}: {sorted(self._obj_map)}"

    __str__ = __repr__
