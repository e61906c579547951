"""Import alias.  The product package lives in ./a-nice-rag_amd (the name the build contract
fixes), which is not a valid Python identifier; `import anrag` resolves to it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "a-nice-rag_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _os, _f, _real
