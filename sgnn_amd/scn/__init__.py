"""`sparseconvnet`-compatible operator namespace backed by hand-written gfx950 HIP kernels.

Drop-in usage for code written against the reference's import (torch/model.py:7):

    import sys, sgnn_amd.scn
    sys.modules['sparseconvnet'] = sgnn_amd.scn      # then `import sparseconvnet as scn` works

Only the names the reference touches plus Deconvolution / BatchNormalization /
NetworkInNetwork are provided.
"""
from .metadata import Metadata, Grid, runtime
from .modules import (InputLayer, OutputLayer, SubmanifoldConvolution, Convolution, Deconvolution, UnPooling,
                      BatchNormReLU, BatchNormalization, FullyConvolutionalNet, Sequential, ConcatTable, AddTable,
                      JoinTable, Identity, SparseToDense, SparseConvNetTensor, NetworkInNetwork)

__all__ = ['InputLayer', 'OutputLayer', 'SubmanifoldConvolution', 'Convolution', 'Deconvolution', 'UnPooling',
           'BatchNormReLU', 'BatchNormalization', 'FullyConvolutionalNet', 'Sequential', 'ConcatTable', 'AddTable',
           'JoinTable', 'Identity', 'SparseToDense', 'SparseConvNetTensor', 'Metadata', 'NetworkInNetwork']
