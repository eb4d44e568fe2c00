"""Import-name shim: `import viai_amd` -> the package in ./vision-infused-audio-inpainter-viai_amd/."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "vision-infused-audio-inpainter-viai_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
