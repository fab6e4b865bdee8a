// Device-side image preprocessing of the VAE input pipeline: what the reference's per-sample
// MONAI transform chain  Resize(patch_size) -> LocalNormalizeByMask -> float32  (data/dataloaders.py:319-329,
// data/transforms.py:8-32) does on CPU worker processes, as two launches over a whole batch of raw images that a
// copy stream has just put in HBM.
//   1. area resize (torch F.interpolate(mode="area") == adaptive average pooling: output (i, j) averages source rows
//      [floor(i*H/Hp), ceil((i+1)*H/Hp)) x the same for columns) + per-image {count, sum, sum of squares} of the
//      NON-ZERO resized pixels (fp64 atomics, a few hundred per image);
//   2. (x - mean) / std on the non-zero pixels (population std; 1.0 when std <= 1e-5), zero stays zero.
// Source images may have different sizes: a small descriptor table gives each one's offset and (H, W).
#include "pti_common.h"

namespace {

struct PreArgs {
  const float* src;          // all raw images of the batch, concatenated (row-major fp32)
  const long long* offset;   // [B] element offset of image b in src
  const int* hw;             // [B][2] source height, width
  float* out;                // [B][1][Hp][Wp] fp32
  double* stats;             // [B][3] count, sum, sumsq (zeroed by the first launch's caller)
  int B, Hp, Wp;
};

__global__ __launch_bounds__(256) void pre_resize_kernel(PreArgs a) {
  const int b = blockIdx.y;
  const int H = a.hw[2 * b], W = a.hw[2 * b + 1];
  const float* img = a.src + a.offset[b];
  const int npix = a.Hp * a.Wp;
  double cnt = 0.0, s1 = 0.0, s2 = 0.0;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
    const int i = p / a.Wp, j = p - i * a.Wp;
    const int y0 = (int)(((long long)i * H) / a.Hp), y1 = (int)((((long long)(i + 1)) * H + a.Hp - 1) / a.Hp);
    const int x0 = (int)(((long long)j * W) / a.Wp), x1 = (int)((((long long)(j + 1)) * W + a.Wp - 1) / a.Wp);
    float acc = 0.f;
    for (int y = y0; y < y1; ++y) {
      const float* row = img + (size_t)y * W;
      for (int x = x0; x < x1; ++x) acc += row[x];
    }
    const float v = acc / (float)((y1 - y0) * (x1 - x0));
    a.out[(size_t)b * npix + p] = v;
    if (v != 0.f) {
      cnt += 1.0;
      s1 += (double)v;
      s2 += (double)v * (double)v;
    }
  }
  // block reduction (wave shuffles, then LDS), one fp64 atomic triple per block
  __shared__ double red[3][4];
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = cnt; red[1][wave] = s1; red[2][wave] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    atomicAdd(&a.stats[3 * b + threadIdx.x], v);
  }
}

__global__ __launch_bounds__(256) void pre_normalize_kernel(PreArgs a) {
  const int b = blockIdx.y;
  const int npix = a.Hp * a.Wp;
  const double cnt = a.stats[3 * b], s1 = a.stats[3 * b + 1], s2 = a.stats[3 * b + 2];
  // an all-zero image: numpy's mean of an empty selection is NaN and (x - NaN)/1 stays NaN except where masked to 0;
  // every pixel is masked then, so the result is all zeros
  const double mean = cnt > 0.0 ? s1 / cnt : 0.0;
  double var = cnt > 0.0 ? s2 / cnt - mean * mean : 0.0;
  if (var < 0.0) var = 0.0;
  const double sd = sqrt(var);
  const float m = (float)mean, inv = sd > 1e-5 ? (float)(1.0 / sd) : 1.0f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
    float* q = a.out + (size_t)b * npix + p;
    const float v = *q;
    *q = (v != 0.f) ? (v - m) * inv : 0.f;
  }
}

}  // namespace

extern "C" int pti_preprocess_batch(const float* src, const int64_t* offsets, const int32_t* hw, int b, int hp, int wp,
                                    float* out, double* stats, pti_stream_t s) {
  if (!src || !offsets || !hw || !out || !stats) PTI_FAIL(PTI_EINVAL, "preprocess_batch: null pointer");
  if (b <= 0 || hp <= 0 || wp <= 0) PTI_FAIL(PTI_EINVAL, "preprocess_batch: bad dims");
  PreArgs a{src, (const long long*)offsets, hw, out, stats, b, hp, wp};
  int bx = (hp * wp + 1023) / 1024;   // ~4 output pixels per thread
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  hipStream_t st = (hipStream_t)s;
  if (hipMemsetAsync(stats, 0, sizeof(double) * 3 * b, st) != hipSuccess) PTI_FAIL(PTI_ELAUNCH, "preprocess_batch: memset failed");
  PTI_LAUNCH(pre_resize_kernel, dim3(bx, b), dim3(256), 0, st, a);
  PTI_CHECK_LAUNCH("preprocess_resize");
  PTI_LAUNCH(pre_normalize_kernel, dim3(bx, b), dim3(256), 0, st, a);
  PTI_CHECK_LAUNCH("preprocess_normalize");
  return PTI_OK;
}
