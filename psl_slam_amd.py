"""Import shim: `import psl_slam_amd` loads the package kept in the directory `psl-slam_amd/`
(the project name has a hyphen, which Python cannot import directly)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "psl-slam_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"), globals())
