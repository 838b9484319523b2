"""get_amd -- MI355X (gfx950) native implementation of the CRIPAC-DIG/GET hot path.

Graph build, gated graph cells with GSL top-k refinement and the two-level concat attention run
as hand-written HIP kernels behind a C-ABI (include/get_hip.h, get_amd/csrc); this package is the
thin host side: ctypes binding, autograd wrappers and drop-in ``nn.Module`` classes with the
reference's names and signatures.  Importing the package does not load the library; the first
device call does, and fails loudly if it has not been built.
"""
from .keywords import KeyWordSettings  # noqa: F401

__all__ = ["KeyWordSettings", "install"]


def install():
    """Make the reference's import paths resolve to this implementation, so that an unmodified
    ``MasterFC/master_get.py`` builds the HIP-backed model:

        import get_amd; get_amd.install()      # before master_get imports its model modules
    """
    from .install import install as _install
    return _install()
