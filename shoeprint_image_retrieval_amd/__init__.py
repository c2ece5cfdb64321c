"""Import alias for the product directory ``shoeprint-image-retrieval_amd/``.

A hyphen cannot appear in a Python module name, so this one-file package only
extends its own ``__path__`` to the product directory: every submodule
(``similarity``, ``network``, ``parse_results``, ``config``, ``_lib`` ...) is the
file of that name under ``shoeprint-image-retrieval_amd/``.
"""

import os as _os

_PRODUCT_DIR = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
    "shoeprint-image-retrieval_amd",
)
__path__.append(_PRODUCT_DIR)

__version__ = "0.1.0"
