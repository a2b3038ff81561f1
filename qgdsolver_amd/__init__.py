"""qgdsolver_amd -- MI355X-native face-flux path of QGDFoam (unicfdlab/QGDsolver).

The product is ``libqgd_amd.so`` (hand-written HIP kernels behind the C-ABI of
``include/qgd_amd.h``); this package is the thin host-side mirror of the
reference's ``fvsc`` / ``QGDThermo`` / QGDFoam-loop interface used by the tests
and the benchmark.  Importing it without the built library raises: there is no
CPU fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is missing)
from ._lib import CaseOptions, QgdError  # noqa: F401
from .mesh import PolyMesh  # noqa: F401
from .fvsc import Device, deviceVolField, fvscStencil, volField  # noqa: F401
from . import fvsc  # noqa: F401
from .qgdfoam import QGDFoamCase, QGDThermo, default_options  # noqa: F401
from . import qhdfoam  # noqa: F401
from . import foamfile  # noqa: F401


def device_count():
    return _lib.lib.qgd_device_count()
