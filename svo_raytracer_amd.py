"""Import shim: the package directory is named ``svo-raytracer_amd`` (not a valid
Python identifier), so ``import svo_raytracer_amd`` resolves here and this module
turns itself into a package whose __path__ is that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "svo-raytracer_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
