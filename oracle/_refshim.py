"""TEST INFRASTRUCTURE ONLY -- import shim for the upstream reference (container only).

Makes ``/root/reference`` importable on a CPU-only host without editing it:
stubs the third-party packages the hot path never touches, and monkey-patches
three source-level incompatibilities.  No-op (raises
``RuntimeError``) when ``/root/reference`` is absent, e.g. on the GPU box.
Nothing here is copied from the reference; it only arranges for it to import.
"""
import importlib.abc
import importlib.machinery
import os
import sys
from unittest import mock

REF = "/root/reference"
_STUB_ROOTS = {"nltk", "hyperopt", "torchvision", "keras", "allennlp", "tensorboardX", "tensorflow",
               "boto3", "botocore", "sacremoses", "ftfy", "spacy", "jieba"}


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__path__ = []
        m.__name__ = spec.name
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        pass


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "Models"))


_done = False


def install():
    """Put the reference on sys.path with the stubs and compat patches in place."""
    global _done
    if not available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    if _done:
        return
    sys.dont_write_bytecode = True
    sys.meta_path.append(_StubFinder())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import numpy as np
    import scipy.sparse as sp
    import torch
    if not hasattr(np, "float"):
        np.float = float                                    # handlers/mz_sampler.py:147
    for cls in (sp.csc_matrix, sp.csr_matrix, sp.coo_matrix):
        if not hasattr(cls, "A"):
            cls.A = property(lambda s: s.toarray())         # interactions.py:18
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self      # Models/BiDAF/wrapper.py:221
    _done = True
