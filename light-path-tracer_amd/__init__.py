"""light-path-tracer_amd: MI355X-native backend of the Light-path-tracer ray integrator.

The modules are flat, like the reference's script directory (`import metrics`, `import image_lens`):
put this directory on sys.path, or import this package, which does so."""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
if _here not in _sys.path:
    _sys.path.insert(0, _here)
