// C-ABI shim around the REFERENCE's own CPU rulebook generators and CPU gather /
// scatter-add functors.  TEST INFRASTRUCTURE ONLY (built into oracle/_ref/, never shipped
// in the product, never copied: the reference sources are compiled where they lie).
//
//   #include <spconv/geometry.h>    /root/reference/mmdet3d/ops/spconv/include/spconv/geometry.h
//        getIndicePairsSubM   :247-297     getIndicePairsConv :144-193
//        getIndicePairsDeConv :195-245
//   src/reordering.cc (compiled as its own translation unit by oracle/Makefile)
//        SparseGatherFunctor<tv::CPU>, SparseScatterAddFunctor<tv::CPU>   :21-50
//
// tensorview.h:16 includes <cuda_runtime_api.h>; the genuine NVIDIA header ships in this
// image with Triton (triton/backends/nvidia/include) and is put on the include path by the
// Makefile -- no stand-in header is written.
//
// The buffers are prepared exactly as the reference's host code does before calling the
// functors (spconv_ops.h:56-62: pairs filled with -1, counts with 0, grid with -1;
// :66-83: sub-manifold forces stride 1 and padding k/2).
#include <spconv/geometry.h>
#include <spconv/reordering.h>

#include <cstdint>
#include <vector>

extern "C" {

// -> number of active outputs (= n); pairs [kvol,2,n] int32, num [kvol] int32
int64_t ref_subm_rulebook(const int *indices, int64_t n, int batch, const int *shape,
                          const int *ksize, const int *dilation, int *pairs, int *num) {
  int kvol = ksize[0] * ksize[1] * ksize[2];
  int64_t vol = (int64_t)shape[0] * shape[1] * shape[2];
  std::vector<int> grid((size_t)batch * vol, -1);
  for (int64_t i = 0; i < (int64_t)kvol * 2 * n; ++i) pairs[i] = -1;
  for (int i = 0; i < kvol; ++i) num[i] = 0;
  int stride[3] = {1, 1, 1};
  int padding[3] = {ksize[0] / 2, ksize[1] / 2, ksize[2] / 2};
  return spconv::getIndicePairsSubM<int, int, 3>(
      tv::TensorView<const int>(indices, {(int)n, 4}),
      tv::TensorView<int>(grid.data(), {(int)(batch * vol)}),
      tv::TensorView<int>(pairs, {kvol, 2, (int)n}), tv::TensorView<int>(num, {kvol}), ksize,
      stride, padding, dilation, shape);
}

// -> number of active outputs; out_indices [n*kvol,4]
int64_t ref_conv_rulebook(const int *indices, int64_t n, int batch, const int *out_shape,
                          const int *ksize, const int *stride, const int *padding,
                          const int *dilation, int transpose, int *out_indices, int *pairs,
                          int *num) {
  int kvol = ksize[0] * ksize[1] * ksize[2];
  int64_t vol = (int64_t)out_shape[0] * out_shape[1] * out_shape[2];
  std::vector<int> grid((size_t)batch * vol, -1);
  for (int64_t i = 0; i < (int64_t)kvol * 2 * n; ++i) pairs[i] = -1;
  for (int i = 0; i < kvol; ++i) num[i] = 0;
  for (int64_t i = 0; i < n * kvol * 4; ++i) out_indices[i] = 0;
  tv::TensorView<const int> in(indices, {(int)n, 4});
  tv::TensorView<int> outi(out_indices, {(int)(n * kvol), 4});
  tv::TensorView<int> g(grid.data(), {(int)(batch * vol)});
  tv::TensorView<int> p(pairs, {kvol, 2, (int)n});
  tv::TensorView<int> c(num, {kvol});
  if (transpose)
    return spconv::getIndicePairsDeConv<int, int, 3>(in, outi, g, p, c, ksize, stride, padding,
                                                     dilation, out_shape);
  return spconv::getIndicePairsConv<int, int, 3>(in, outi, g, p, c, ksize, stride, padding,
                                                 dilation, out_shape);
}

// the reference's CPU row gather: buffer[i,:] = features[idx[i],:]
void ref_sparse_gather_f32(float *buffer, const float *features, int64_t nrows, int planes,
                           const int *idx, int size) {
  spconv::functor::SparseGatherFunctor<tv::CPU, float, int> f;
  f(tv::CPU(), tv::TensorView<float>(buffer, {size, planes}),
    tv::TensorView<const float>(features, {(int)nrows, planes}),
    tv::TensorView<const int>(idx, {size}), size);
}

// the reference's CPU scatter-add: out[idx[i],:] += buffer[i,:]
void ref_sparse_scatter_add_f32(float *out, int64_t nrows, int planes, const float *buffer,
                                const int *idx, int size) {
  spconv::functor::SparseScatterAddFunctor<tv::CPU, float, int> f;
  f(tv::CPU(), tv::TensorView<float>(out, {(int)nrows, planes}),
    tv::TensorView<const float>(buffer, {size, planes}), tv::TensorView<const int>(idx, {size}),
    size, true);
}
}
