"""Regular (strided) and transposed sparse-conv rulebooks -- the non sub-manifold branch of
spconv::getIndicePair (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:105-141; CPU functor
geometry.h:144-245, GPU kernels indice.cu.h:22-145).  Kernel: ococc_conv_rulebook_build."""
import numpy as np
import torch

from .. import _lib as L


def build_regular_rulebook(indices, batch_size, out_shape, ksize, stride, padding, dilation, transpose):
    """-> (outids [M,4] int32 sorted by (b,z,y,x), indice_pairs [K,2,N] int32, indice_pair_num [K])."""
    n = indices.size(0)
    kvol = int(np.prod(ksize))
    dev = indices.device
    cells = int(batch_size) * int(np.prod(out_shape))
    cap = max(min(n * kvol, cells), 1)
    nbytes = L.lib.ococc_conv_rulebook_workspace_bytes(n, int(batch_size), L.i3(out_shape), L.i3(ksize))
    if nbytes < 0:
        raise L.OcoccError('get_indice_pairs: unsupported geometry (batch*D*H*W < 2^31 required)')
    ws = L.workspace(nbytes, dev)
    outids = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    pairs = torch.empty((kvol, 2, n), dtype=torch.int32, device=dev)
    num = torch.empty((kvol,), dtype=torch.int32, device=dev)
    num_out = torch.zeros((1,), dtype=torch.int32, device=dev)
    L.check(L.lib.ococc_conv_rulebook_build(L.ptr(indices), n, int(batch_size), L.i3(out_shape), L.i3(ksize),
                                            L.i3(stride), L.i3(padding), L.i3(dilation), int(bool(transpose)),
                                            L.ptr(outids), cap, L.ptr(pairs), L.ptr(num), L.ptr(num_out),
                                            L.ptr(ws), ws.numel(), L.stream()), 'conv_rulebook_build')
    m = int(num_out.item())  # the reference reads numActOut back as well (spconv_ops.h:131-137)
    return outids[:m], pairs, num
