"""Import shim: the package directory is named `draco-sharp_amd/` (not a valid
Python identifier), so this module loads it under the importable name
`draco_sharp_amd`.  `import draco_sharp_amd.synth` etc. resolve inside that
directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "draco-sharp_amd")]
__package__ = "draco_sharp_amd"
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
