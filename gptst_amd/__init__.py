"""Import alias: the product package lives in ``gpt-st_amd/`` (a name Python cannot
import directly because of the hyphen).  ``import gptst_amd`` resolves every submodule
from that directory, so ``gptst_amd.model`` is ``gpt-st_amd/model.py``."""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
__path__ = [_os.path.join(_os.path.dirname(_here), "gpt-st_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f, _here, _os
