"""Regular (strided) and transposed sparse-conv rulebooks -- the non sub-manifold branch of
spconv::getIndicePair (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:105-141; CPU functor
geometry.h:144-245, GPU kernels indice.cu.h:22-145)."""


def build_regular_rulebook(indices, batch_size, out_shape, ksize, stride, padding, dilation,
                           transpose):
    raise NotImplementedError(
        'SparseConv3d / SparseInverseConv3d rulebooks are scheduled after the sub-manifold path '
        '(SURVEY.md section 8 row B3); OcOccNet and the benchmark configs use SubMConv3d only')
