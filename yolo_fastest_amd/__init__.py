"""Importable alias of the package directory `yolo-fastest-and-embedded-deployment_amd/` (a hyphenated name cannot
be written in an `import` statement): `import yolo_fastest_amd` loads that directory as this package."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "yolo-fastest-and-embedded-deployment_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
