// SURVEY 8(f) row 4 -- the visibility ray test of the GT-occupancy annotation
// (tools/occ/occ_annotate.py: point_cloud_to_range_image_idx :141-207 and its use in
// OccAnnotator.annotate_trk :488-556).  For every unoccupied cell centre of an object grid, every
// frame of the tracklet and every LiDAR: move the centre into the sensor frame, find its range-image
// pixel (row = beam with the nearest inclination, column from the azimuth) and compare the measured
// range with the centre's: a return at or behind the centre means the ray passed through the cell,
// i.e. the cell is EMPTY (label 2); a cell no ray ever crossed stays unknown (0).
//
// The reference does this with a dozen full-size [B,N,*] float64 torch temporaries per sensor
// (einsum, norm, atan2, an [N,H] difference matrix per frame for the argmin, gathers, max over
// frames); here it is one pass: thread = (sensor-frame, centre), float64 arithmetic as in the
// reference, the H beam inclinations of the sensor-frame in LDS, the verdict OR-ed into the centre's
// label with a plain store (every writer stores the same 2).  HBM-bound on the centres:
// N*24 B in, N*4 B out per object, the 2-5 MB range images are gathered (L2 resident).
#include "common.hpp"

namespace {

struct Affine {  // y = R x + t, row-major R
  double r[9], t[3];
};

constexpr int kMaxBeams = 256;

__global__ void __launch_bounds__(256)
visibility_kernel(const double* __restrict__ centers, int64_t n, const Affine* __restrict__ to_ego,
                  const Affine* __restrict__ to_sensor, const double* __restrict__ az_corr,
                  const double* __restrict__ inclinations, int height, int width, const void* const* __restrict__ range_images,
                  int range_is_f64, int frames, int32_t* __restrict__ vis, int32_t* __restrict__ dbg_idx,
                  double* __restrict__ dbg_range) {
  // blockIdx.y = sensor-frame index sf (sensor-major: sf = sensor * frames + frame)
  __shared__ double inc[kMaxBeams];
  const int sf = blockIdx.y;
  for (int h = threadIdx.x; h < height; h += blockDim.x) inc[h] = inclinations[(int64_t)sf * height + h];
  __syncthreads();
  const Affine a = to_ego[sf % frames], s = to_sensor[sf];
  const double azc = az_corr[sf];
  const float two_pi_f = 6.2831854820251465f;  // the reference multiplies a float32 mask by 2*pi: float32 constant
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double lx = centers[i * 3], ly = centers[i * 3 + 1], lz = centers[i * 3 + 2];
    const double ex = a.r[0] * lx + a.r[1] * ly + a.r[2] * lz + a.t[0];
    const double ey = a.r[3] * lx + a.r[4] * ly + a.r[5] * lz + a.t[1];
    const double ez = a.r[6] * lx + a.r[7] * ly + a.r[8] * lz + a.t[2];
    const double px = s.r[0] * ex + s.r[1] * ey + s.r[2] * ez + s.t[0];
    const double py = s.r[3] * ex + s.r[4] * ey + s.r[5] * ez + s.t[1];
    const double pz = s.r[6] * ex + s.r[7] * ey + s.r[8] * ez + s.t[2];
    const double xy = sqrt(px * px + py * py);
    const double pinc = atan2(pz, xy);
    int row = 0;
    double best = fabs(pinc - inc[0]);
    for (int h = 1; h < height; ++h) {  // first minimum, as torch.argmin
      const double d = fabs(pinc - inc[h]);
      if (d < best) {
        best = d;
        row = h;
      }
    }
    double az = atan2(py, px) + azc;
    if (az > 3.141592653589793) az -= (double)two_pi_f;
    else if (az < -3.141592653589793) az += (double)two_pi_f;
    double colf = (double)width - 1.0 + 0.5 - (az + 3.141592653589793) / (2.0 * 3.141592653589793) * (double)width;
    colf = rint(colf);                 // torch.round: half to even
    const int col = (int)fmod(colf, (double)width);
    const double range = sqrt(px * px + py * py + pz * pz);
    if (dbg_idx) {
      dbg_idx[((int64_t)sf * n + i) * 2] = row;
      dbg_idx[((int64_t)sf * n + i) * 2 + 1] = col;
      dbg_range[(int64_t)sf * n + i] = range;
    }
    if (vis && col >= 0 && col < width) {
      const int64_t pix = (int64_t)row * width + col;
      const double measured = range_is_f64 ? ((const double*)range_images[sf])[pix] : (double)((const float*)range_images[sf])[pix];
      if (measured >= range) vis[i] = 2;  // benign race: every writer stores 2
    }
  }
}

}  // namespace

extern "C" int ococc_occ_visibility_f64(const double* centers, int64_t n, const double* to_ego, int32_t frames,
                                        const double* to_sensor, const double* az_corr, const double* inclinations,
                                        int32_t sensors, int32_t height, int32_t width, const void* const* range_images,
                                        int32_t range_dtype, int32_t* visibility, int32_t* dbg_indices,
                                        double* dbg_range, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(n >= 0 && frames >= 1 && sensors >= 1, "bad sizes");
  OCOCC_REQUIRE(height >= 1 && height <= kMaxBeams && width >= 1, "range image height must be 1..256");
  OCOCC_REQUIRE(range_dtype == OCOCC_F32 || range_dtype == 2, "range_dtype: 0 = f32, 2 = f64");
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(centers && to_ego && to_sensor && az_corr && inclinations, "null pointer");
  OCOCC_REQUIRE(visibility == nullptr || range_images != nullptr, "visibility needs the range images");
  OCOCC_REQUIRE((dbg_indices == nullptr) == (dbg_range == nullptr), "dbg_indices and dbg_range go together");
  if (visibility) OCOCC_HIP(hipMemsetAsync(visibility, 0, n * sizeof(int32_t), stream));
  const dim3 grid((unsigned)ococc_grid_1d(n, 256, 1024), (unsigned)(sensors * frames));
  hipLaunchKernelGGL(visibility_kernel, grid, dim3(256), 0, stream, centers, n, (const Affine*)to_ego,
                     (const Affine*)to_sensor, az_corr, inclinations, (int)height, (int)width, range_images,
                     (int)(range_dtype == 2), (int)frames, visibility, dbg_indices, dbg_range);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
