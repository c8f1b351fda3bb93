"""Import shim: the real package lives in ``is-fusion_amd/`` (a directory name Python cannot import).

``import isfusion_amd`` executes ``is-fusion_amd/__init__.py`` with this package's ``__path__`` pointing
there, so ``isfusion_amd.spconv`` etc. resolve to ``is-fusion_amd/spconv.py``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "is-fusion_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
