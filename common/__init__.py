"""Drop-in import path for the reference's second model family (`common.myhand.lijun_model_graph`), see
common/myhand/."""

import os as _os
import sys as _sys

# Keep the reference's own sub-modules of this package importable when its checkout is ALSO on sys.path (behind this
# repository): a regular package shadows same-named directories further down the path, so they are appended to
# __path__ here -- modules defined in this directory win, everything else resolves to the reference.
for _p in list(_sys.path):
    _cand = _os.path.join(_p or '.', *__name__.split('.'))
    if _os.path.isdir(_cand) and _os.path.abspath(_cand) != _os.path.dirname(_os.path.abspath(__file__)) \
            and _cand not in __path__:
        __path__.append(_cand)
