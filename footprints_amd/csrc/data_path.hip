// Device-side data path: what the reference does per sample on DataLoader workers between the file
// readers and the network -- horizontal flip, torchvision ColorJitter (through Pillow), ToTensor, and the label algebra of
// KITTIDataset / MatterportDataset.__getitem__ -- as two kernels over a whole batch that arrived by ONE pinned H2D copy:
//   reference: footprints/datasets/footprint_dataset.py:55-65,73-75,84-85; kitti_dataset.py:66-67,72-73,86-97,105-112;
//              matterport_dataset.py:69-78,93-97; footprints/utils.py:27-33.
// Byte work is bit-exact with Pillow 12 (oracle/data_path.py restates libImaging's Blend.c / Convert.c arithmetic and is checked
// against the real library over all 2^24 colours): Image.blend in float32 with truncation, the L24 fixed-point luma, rgb <-> hsv
// with libImaging's mix of float variables and double literals, C round().  This file is compiled with -ffp-contract=off: a fused
// multiply-add anywhere in these expressions would change bytes.
#include "fp_common.h"

namespace {

struct AugParams {          // one per sample, uploaded with the batch (fp_aug_params in the header)
  int32_t flip;             // horizontal flip of the image and of every label map
  int32_t n_ops;            // 0 = no colour jitter, else 4
  int32_t ops[4];           // application order: 0 brightness, 1 contrast, 2 saturation, 3 hue
  float factor[4];          // indexed by op id (the hue entry is unused: hue_shift carries np.uint8(hue_factor * 255))
  int32_t hue_shift;
  int32_t pad;
};

__device__ __forceinline__ int blend_byte(int in1, int in2, float alpha) {        // libImaging/Blend.c
  const float t = __fadd_rn((float)in1, __fmul_rn(alpha, (float)(in2 - in1)));
  if (alpha >= 0.f && alpha <= 1.f) return ((int)t) & 0xff;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }   // Convert.c L24

__device__ __forceinline__ void rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {       // Convert.c rgb2hsv_row
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) { uh = 0; us = 0; return; }
  const float cr = (float)(maxc - minc);
  const float s = __fdiv_rn(cr, (float)maxc);
  const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
  float h;
  if (r == maxc) h = __fsub_rn(bc, gc);
  else if (g == maxc) h = (float)__dsub_rn(__dadd_rn(2.0, (double)rc), (double)bc);
  else h = (float)__dsub_rn(__dadd_rn(4.0, (double)gc), (double)rc);
  h = (float)fmod(__dadd_rn(__ddiv_rn((double)h, 6.0), 1.0), 1.0);
  uh = min(max((int)__dmul_rn((double)h, 255.0), 0), 255);
  us = min(max((int)__dmul_rn((double)s, 255.0), 0), 255);
}

__device__ __forceinline__ void hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {          // Convert.c hsv2rgb
  if (s == 0) { r = g = b = v; return; }
  const double hd = __ddiv_rn(__dmul_rn((double)(float)h, 6.0), 255.0);
  const int i = (int)floor(hd);
  const double f = (double)(float)__dsub_rn(hd, (double)(float)i);
  const double fs = (double)(float)__ddiv_rn((double)(float)s, 255.0);
  const double vf = (double)(float)v;
  const int p = (int)round(__dmul_rn(vf, __dsub_rn(1.0, fs)));
  const int q = (int)round(__dmul_rn(vf, __dsub_rn(1.0, __dmul_rn(fs, f))));
  const int t = (int)round(__dmul_rn(vf, __dsub_rn(1.0, __dmul_rn(fs, __dsub_rn(1.0, f)))));
  const int up = min(max(p, 0), 255), uq = min(max(q, 0), 255), ut = min(max(t, 0), 255);
  switch (i % 6) {
    case 0: r = v; g = ut; b = up; break;
    case 1: r = uq; g = v; b = up; break;
    case 2: r = up; g = v; b = ut; break;
    case 3: r = up; g = uq; b = v; break;
    case 4: r = ut; g = up; b = v; break;
    default: r = v; g = up; b = uq; break;
  }
}

// the jitter ops [first, last) of one sample on one pixel; `mean` = the contrast op's grey level (valid once its pass has run)
__device__ __forceinline__ void apply_ops(const AugParams& p, int first, int last, int mean, int& r, int& g, int& b) {
  for (int k = first; k < last; ++k) {
    const int op = p.ops[k];
    if (op == 0) {                                     // ImageEnhance.Brightness: blend(black, image, factor)
      const float f = p.factor[0];
      r = blend_byte(0, r, f); g = blend_byte(0, g, f); b = blend_byte(0, b, f);
    } else if (op == 1) {                              // ImageEnhance.Contrast: blend(mean grey, image, factor)
      const float f = p.factor[1];
      r = blend_byte(mean, r, f); g = blend_byte(mean, g, f); b = blend_byte(mean, b, f);
    } else if (op == 2) {                              // ImageEnhance.Color: blend(image.convert("L"), image, factor)
      const float f = p.factor[2];
      const int l = luma(r, g, b);
      r = blend_byte(l, r, f); g = blend_byte(l, g, f); b = blend_byte(l, b, f);
    } else {                                           // adjust_hue: HSV, h += np.uint8(hue_factor * 255) (wraps), back to RGB
      int h, s, v;
      rgb2hsv(r, g, b, h, s, v);
      h = (h + p.hue_shift) & 0xff;
      hsv2rgb(h, s, v, r, g, b);
    }
  }
}

__device__ __forceinline__ int contrast_pos(const AugParams& p) {
  for (int k = 0; k < p.n_ops; ++k)
    if (p.ops[k] == 1) return k;
  return -1;
}

// pass 1: sum of the luma of every pixel as the image stands right before its contrast op (integers: atomics are exact and
// order-independent).  grid = (blocks per image, batch)
__global__ void __launch_bounds__(256) jitter_luma_sum_kernel(const unsigned char* __restrict__ img, const AugParams* __restrict__ params, int HW,
                                                              unsigned long long* __restrict__ sums) {
  const AugParams p = params[blockIdx.y];
  const int cp = contrast_pos(p);
  if (cp < 0) return;
  const unsigned char* im = img + (size_t)blockIdx.y * HW * 3;
  unsigned long long s = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    int r = im[i * 3], g = im[i * 3 + 1], b = im[i * 3 + 2];
    apply_ops(p, 0, cp, 0, r, g, b);
    s += (unsigned)luma(r, g, b);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ unsigned long long sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&sums[blockIdx.y], sm[0] + sm[1] + sm[2] + sm[3]);
}

// pass 2: flip + every jitter op + ToTensor: uint8 [B][H][W][3] -> float [B][3][H][W] in [0, 1]
__global__ void __launch_bounds__(256) assemble_image_kernel(const unsigned char* __restrict__ img, const AugParams* __restrict__ params,
                                                             const unsigned long long* __restrict__ sums, float* __restrict__ out, int H, int W) {
  const AugParams p = params[blockIdx.y];
  const int HW = H * W;
  // int(ImageStat.Stat(image.convert("L")).mean[0] + 0.5): the mean is sum / count in double
  const int mean = p.n_ops ? (int)(__dadd_rn(__ddiv_rn((double)sums[blockIdx.y], (double)HW), 0.5)) : 0;
  const unsigned char* im = img + (size_t)blockIdx.y * HW * 3;
  float* o = out + (size_t)blockIdx.y * 3 * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const int y = i / W, x = i - y * W;
    const int sx = p.flip ? W - 1 - x : x;
    const unsigned char* px = im + ((size_t)y * W + sx) * 3;
    int r = px[0], g = px[1], b = px[2];
    apply_ops(p, 0, p.n_ops, mean, r, g, b);
    o[i] = __fdiv_rn((float)r, 255.f);
    o[HW + i] = __fdiv_rn((float)g, 255.f);
    o[2 * HW + i] = __fdiv_rn((float)b, 255.f);
  }
}

struct LabelArgs {
  const void *vg, *gd, *dm, *aux, *mov;      // [B][H][W] maps of type T as they leave the resize (NOT flipped); aux = disparity (KITTI) / raw depth (Matterport)
  const AugParams* params;
  float *o_vg, *o_depth, *o_gd, *o_mov, *o_dm, *o_ag;
  int H, W;
  int dataset;                               // 0 KITTI, 1 Matterport
  int no_depth_mask, project_down_baseline, use_moving;
  double threshold, fxb, depth_scaling;
};

// label algebra in float64 like numpy in the reference, one cast to float32 at the end (torch.tensor(val).float())
template <typename T>
__global__ void __launch_bounds__(256) assemble_labels_kernel(const LabelArgs a) {
  const int HW = a.H * a.W;
  const AugParams p = a.params[blockIdx.y];
  const size_t base = (size_t)blockIdx.y * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const int y = i / a.W, x = i - y * a.W;
    const size_t s = base + (size_t)y * a.W + (p.flip ? a.W - 1 - x : x);
    const double vgp = (double)reinterpret_cast<const T*>(a.vg)[s];
    double gd = (double)reinterpret_cast<const T*>(a.gd)[s];
    double dm = (double)reinterpret_cast<const T*>(a.dm)[s];
    const double aux = (double)reinterpret_cast<const T*>(a.aux)[s];
    const double vg = vgp > a.threshold ? 1.0 : 0.0;                    // kitti_dataset.py:67 / matterport_dataset.py:60
    double depth, mov;
    if (a.dataset == 0) {
      if (a.project_down_baseline) gd = 1.0;                            // kitti_dataset.py:72-73
      if (a.no_depth_mask) dm = __dmul_rn(dm, 0.0);                     // :86-87
      if (dm != 0.0) gd = 0.0;                                          // :90
      const double disp = __dsub_rn(aux, 1.25);                         // :94-96
      depth = __ddiv_rn(a.fxb, __dsub_rn(disp, disp == 0.0 ? 1.0 : 0.0));   // utils.py:31
      if (depth < 0.0) depth = 0.0;                                     // utils.py:32
      mov = a.use_moving ? (double)reinterpret_cast<const T*>(a.mov)[s] : 0.0;   // kitti_dataset.py:99-103
      mov = __dmul_rn(mov, __dsub_rn(1.0, vg));                         // :106
      mov = __dmul_rn(mov, __dsub_rn(1.0, dm));                         // :108
    } else {
      depth = __dmul_rn(aux, a.depth_scaling);                          // matterport_dataset.py:70
      if (gd == 0.1) gd = 0.0;                                          // :73
      gd = __dmul_rn(gd, gd < 10.0 ? 1.0 : 0.0);                        // :76
      mov = 0.0;                                                        // :79
      if (a.no_depth_mask) dm = __dmul_rn(dm, 0.0);                     // :93-94
      if (dm != 0.0) gd = 0.0;                                          // :97
    }
    const size_t o = base + i;
    const float fvg = (float)vg, fgd = (float)gd;
    a.o_vg[o] = fvg;
    a.o_depth[o] = (float)depth;
    a.o_gd[o] = fgd;
    a.o_mov[o] = (float)mov;
    a.o_dm[o] = (float)dm;
    a.o_ag[o] = __fadd_rn(fgd, fvg) > 0.f ? 1.f : 0.f;                  // footprint_dataset.py:64
  }
}

}  // namespace

extern "C" int fp_assemble_images(const uint8_t* images_hwc, const void* aug_params, uint64_t* luma_sums, float* out_nchw, int32_t B, int32_t H,
                                  int32_t W, fp_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FP_REQUIRE(images_hwc && aug_params && luma_sums && out_nchw && B > 0 && H > 0 && W > 0, "fp_assemble_images: bad arguments");
  FP_REQUIRE((int64_t)H * W * 255 < ((int64_t)1 << 62), "fp_assemble_images: image too large");
  hipError_t e = hipMemsetAsync(luma_sums, 0, (size_t)B * sizeof(uint64_t), stream);
  if (e != hipSuccess) return fp_set_error((int)e, "fp_assemble_images: %s", hipGetErrorString(e));
  const int HW = H * W;
  int bx = (HW + 255) / 256;
  if (bx > 64) bx = 64;
  fp_launch(jitter_luma_sum_kernel, dim3(bx, B), dim3(256), 0, stream, images_hwc, (const AugParams*)aug_params, HW,
                     (unsigned long long*)luma_sums);
  int rc = fp_check_launch("fp_assemble_images(luma)");
  if (rc) return rc;
  int bx2 = (HW + 255) / 256;
  if (bx2 > 256) bx2 = 256;
  fp_launch(assemble_image_kernel, dim3(bx2, B), dim3(256), 0, stream, images_hwc, (const AugParams*)aug_params,
                     (const unsigned long long*)luma_sums, out_nchw, H, W);
  return fp_check_launch("fp_assemble_images");
}

extern "C" int fp_assemble_labels(const void* visible_ground, const void* ground_depth, const void* depth_mask, const void* aux, const void* moving,
                                  int32_t is_double, const void* aug_params, float* o_visible_ground, float* o_depth, float* o_ground_depth,
                                  float* o_moving, float* o_depth_mask, float* o_all_ground, int32_t B, int32_t H, int32_t W, int32_t dataset,
                                  int32_t no_depth_mask, int32_t project_down_baseline, int32_t use_moving, double threshold, double fxb,
                                  double depth_scaling, fp_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FP_REQUIRE(visible_ground && ground_depth && depth_mask && aux && aug_params && o_visible_ground && o_depth && o_ground_depth && o_moving &&
                 o_depth_mask && o_all_ground && B > 0 && H > 0 && W > 0,
             "fp_assemble_labels: null pointer / bad size");
  FP_REQUIRE(dataset == 0 || dataset == 1, "fp_assemble_labels: dataset must be 0 (KITTI) or 1 (Matterport)");
  FP_REQUIRE(!(dataset == 0 && use_moving) || moving, "fp_assemble_labels: the moving-object map is missing");
  LabelArgs a;
  a.vg = visible_ground; a.gd = ground_depth; a.dm = depth_mask; a.aux = aux; a.mov = moving; a.params = (const AugParams*)aug_params;
  a.o_vg = o_visible_ground; a.o_depth = o_depth; a.o_gd = o_ground_depth; a.o_mov = o_moving; a.o_dm = o_depth_mask; a.o_ag = o_all_ground;
  a.H = H; a.W = W; a.dataset = dataset; a.no_depth_mask = no_depth_mask; a.project_down_baseline = project_down_baseline; a.use_moving = use_moving;
  a.threshold = threshold; a.fxb = fxb; a.depth_scaling = depth_scaling;
  int bx = (H * W + 255) / 256;
  if (bx > 256) bx = 256;
  if (is_double) fp_launch(assemble_labels_kernel<double>, dim3(bx, B), dim3(256), 0, stream, a);
  else fp_launch(assemble_labels_kernel<float>, dim3(bx, B), dim3(256), 0, stream, a);
  return fp_check_launch("fp_assemble_labels");
}

extern "C" int32_t fp_aug_params_bytes(void) { return (int32_t)sizeof(AugParams); }
