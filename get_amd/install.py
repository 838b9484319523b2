"""``sys.modules`` shims: the reference's module paths -> get_amd's drop-in classes."""
import sys
import types


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__get_amd_shim__ = True
    sys.modules[name] = m
    return m


def install():
    import numpy as np
    import torch
    import torch.nn as nn

    from . import modules as M
    from .keywords import KeyWordSettings

    # Models.BiDAF.wrapper  (master_get -> graph_based_semantic_structure.py:8)
    _module("Models.BiDAF.wrapper", GGNN=M.GGNN, GGNN_with_GSL=M.GGNN_with_GSL, GSL=M.GSL, Linear=M.Linear,
            LSTM=M.LSTM, torch=torch, nn=nn)
    # thirdparty.two_branches_attention is star-imported by the model module (:9), which is also
    # where that module gets `nn` and `np` from
    _module("thirdparty.two_branches_attention", ConcatNotEqualSelfAtt=M.ConcatNotEqualSelfAtt,
            ConcatSelfAtt=M.ConcatSelfAtt, torch=torch, nn=nn, np=np)
    _module("thirdparty.self_attention", MultiHeadSelfAttentionICLR2017Extend=M.MultiHeadSelfAttentionICLR2017Extend)
    # the model itself (master_get.py:5,145)
    _module("Models.FCWithEvidences.graph_based_semantic_structure",
            Graph_basedSemantiStructure=M.Graph_basedSemantiStructure, KeyWordSettings=KeyWordSettings)
    return M
