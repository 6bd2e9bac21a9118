// Image pre-processing in front of the tokenizer: PIL-exact antialiased resize of a uint8 RGB image,
// optional centre crop, ToTensor (/255) and CLIP Normalize, written as the [3,S,S] float tensor encode_image consumes.
//
// Replaces, on the device,
//   transforms.Resize((S,S), interpolation=3) -> ToTensor -> Normalize     models/seed_llama_tokenizer.py:50-56 (bicubic)
//   transforms.Resize(S) -> CenterCrop(S) -> ToTensor -> Normalize          models/transforms.py:8-21        (bilinear)
// whose arithmetic lives in third-party code that is not under /root/reference: Pillow's ImagingResample (8 bits per channel:
// separable two-pass convolution, coefficients from a double-precision filter kernel normalised per output pixel and
// converted to 22-bit fixed point, uint8 rounding after EACH pass) and torchvision's ToTensor / Normalize (fp32 x/255,
// (x-mean)/std).  The coefficient tables are built on the host in double precision exactly the way Pillow's
// precompute_coeffs / normalize_coeffs_8bpc do; the device kernels do the integer convolutions and the float epilogue.
// Byte work, HBM/L2-bound: one thread per output pixel (3 channels), windows of neighbouring threads overlap in L1/L2.
#include <math.h>
#include <string.h>
#include <vector>
#include "common.h"
#include "seedmi_internal.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;      // Pillow Resample.c

#pragma clang fp contract(off)
double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}
double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

struct Coeffs {
    int ksize = 0;
    std::vector<int> kk;        // [outSize][ksize] fixed point
    std::vector<int> bounds;    // [outSize][2] = xmin, count
};

// Pillow precompute_coeffs + normalize_coeffs_8bpc for the whole-image box (in0 = 0, in1 = inSize)
void precompute(int inSize, int outSize, int filter, Coeffs& c) {
    const double fsupport = filter == 3 ? 2.0 : 1.0;
    const double scale = (double)inSize / outSize;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = fsupport * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    c.ksize = ksize;
    c.kk.assign((size_t)outSize * ksize, 0);
    c.bounds.assign((size_t)outSize * 2, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < outSize; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > inSize) xmax = inSize;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double arg = (x + xmin - center + 0.5) * ss;
            const double w = filter == 3 ? bicubic_filter(arg) : bilinear_filter(arg);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (int x = 0; x < xmax; ++x) {
            const double v = k[x] * (1 << PRECISION_BITS);
            c.kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
        c.bounds[2 * xx] = xmin;
        c.bounds[2 * xx + 1] = xmax;
    }
}

int grid_for(long long total, int block) { return (int)((total + block - 1) / block); }

SEEDMI_DEVINL int clip8(int v) {                 // Pillow clip8: (in >> PRECISION_BITS) clamped to [0, 255]
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: rows [y0, y0+ny) of the source -> tmp[(y-y0)][xx][3]
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, int src_stride, int y0, int ny,
                                                       int out_w, const int* __restrict__ kk, const int* __restrict__ bounds,
                                                       int ksize, uint8_t* __restrict__ tmp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ny * out_w) return;
    const int y = idx / out_w, xx = idx - y * out_w;
    const int xmin = bounds[2 * xx], xcnt = bounds[2 * xx + 1];
    const int* k = kk + (size_t)xx * ksize;
    const uint8_t* row = src + (size_t)(y0 + y) * src_stride + 3 * xmin;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xcnt; ++x) {
        const int c = k[x];
        s0 += row[3 * x] * c;
        s1 += row[3 * x + 1] * c;
        s2 += row[3 * x + 2] * c;
    }
    uint8_t* o = tmp + ((size_t)y * out_w + xx) * 3;
    o[0] = (uint8_t)clip8(s0);
    o[1] = (uint8_t)clip8(s1);
    o[2] = (uint8_t)clip8(s2);
}

// vertical pass + crop + ToTensor + Normalize: out[c][yy][xx], yy/xx in crop coordinates
template <bool FP32>
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* __restrict__ tmp, int tmp_w, int y0,
                                                            const int* __restrict__ kk, const int* __restrict__ bounds,
                                                            int ksize, int crop_top, int crop_left, int out_h, int out_w,
                                                            float m0, float m1, float m2, float d0, float d1, float d2,
                                                            void* __restrict__ out, uint8_t* __restrict__ out_u8) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= out_h * out_w) return;
    const int yy = idx / out_w, xx = idx - yy * out_w;
    const int ry = yy + crop_top, rx = xx + crop_left;             // coordinates in the resized image
    const int ymin = bounds[2 * ry], ycnt = bounds[2 * ry + 1];
    const int* k = kk + (size_t)ry * ksize;
    const uint8_t* col = tmp + ((size_t)(ymin - y0) * tmp_w + rx) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < ycnt; ++y) {
        const int c = k[y];
        const uint8_t* px = col + (size_t)y * tmp_w * 3;
        s0 += px[0] * c;
        s1 += px[1] * c;
        s2 += px[2] * c;
    }
    const int u0 = clip8(s0), u1 = clip8(s1), u2 = clip8(s2);
    if (out_u8) {
        uint8_t* o = out_u8 + (size_t)idx * 3;
        o[0] = (uint8_t)u0; o[1] = (uint8_t)u1; o[2] = (uint8_t)u2;
    }
    // ToTensor: float(u8) / 255 ; Normalize: (x - mean) / std   (IEEE fp32 division, like torch on the host)
    const float f0 = ((float)u0 / 255.0f - m0) / d0;
    const float f1 = ((float)u1 / 255.0f - m1) / d1;
    const float f2 = ((float)u2 / 255.0f - m2) / d2;
    const size_t plane = (size_t)out_h * out_w;
    if (FP32) {
        float* o = (float*)out;
        o[idx] = f0; o[plane + idx] = f1; o[2 * plane + idx] = f2;
    } else {
        bf16_t* o = (bf16_t*)out;
        o[idx] = f2bf(f0); o[plane + idx] = f2bf(f1); o[2 * plane + idx] = f2bf(f2);
    }
}

}  // namespace

static size_t pre_ws(int in_h, int in_w, int resize_h, int resize_w, int filter, size_t off[5]) {
    auto ks = [&](int in, int out) {
        double fs = (double)in / out;
        if (fs < 1.0) fs = 1.0;
        return (size_t)((int)ceil((filter == 3 ? 2.0 : 1.0) * fs) * 2 + 1);
    };
    size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o += (b + 255) & ~(size_t)255; return r; };
    off[0] = take((size_t)resize_w * ks(in_w, resize_w) * 4);
    off[1] = take((size_t)resize_w * 2 * 4);
    off[2] = take((size_t)resize_h * ks(in_h, resize_h) * 4);
    off[3] = take((size_t)resize_h * 2 * 4);
    off[4] = take((size_t)in_h * resize_w * 3);
    return o;
}

extern "C" size_t seedmi_preprocess_workspace_bytes(int in_h, int in_w, int resize_h, int resize_w, int filter) {
    if (in_h <= 0 || in_w <= 0 || resize_h <= 0 || resize_w <= 0 || (filter != 2 && filter != 3)) return 0;
    size_t off[5];
    return pre_ws(in_h, in_w, resize_h, resize_w, filter, off);
}

extern "C" int seedmi_preprocess_image_u8(const void* rgb_hwc, int in_h, int in_w, int row_stride, int resize_h, int resize_w,
                                          int filter, int crop_top, int crop_left, int out_h, int out_w, const float* mean3,
                                          const float* std3, void* out_chw, int out_is_fp32, void* out_u8_hwc, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    if (!rgb_hwc || !out_chw || !mean3 || !std3 || in_h <= 0 || in_w <= 0 || resize_h <= 0 || resize_w <= 0 ||
        row_stride < 3 * in_w || (filter != 2 && filter != 3) || crop_top < 0 || crop_left < 0 || out_h <= 0 || out_w <= 0 ||
        crop_top + out_h > resize_h || crop_left + out_w > resize_w) {
        seedmi_set_error("seedmi_preprocess_image_u8: bad shape in %dx%d resize %dx%d crop (%d,%d) out %dx%d filter %d", in_h,
                         in_w, resize_h, resize_w, crop_top, crop_left, out_h, out_w, filter);
        return SEEDMI_E_SHAPE;
    }
    size_t off[5];
    const size_t need = pre_ws(in_h, in_w, resize_h, resize_w, filter, off);
    if (!workspace || workspace_bytes < need) {
        seedmi_set_error("seedmi_preprocess_image_u8: workspace too small (%zu < %zu)", workspace_bytes, need);
        return SEEDMI_E_SHAPE;
    }
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    Coeffs ch, cv;
    precompute(in_w, resize_w, filter, ch);
    precompute(in_h, resize_h, filter, cv);
    // pageable-source async copies are staged by the runtime before they return, so the vectors may die with this frame
    hipError_t e = hipMemcpyAsync(ws + off[0], ch.kk.data(), ch.kk.size() * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(ws + off[1], ch.bounds.data(), ch.bounds.size() * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(ws + off[2], cv.kk.data(), cv.kk.size() * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(ws + off[3], cv.bounds.data(), cv.bounds.size() * 4, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) {
        seedmi_set_error("seedmi_preprocess_image_u8: coefficient upload: %s", hipGetErrorString(e));
        return SEEDMI_E_HIP;
    }
    // only the source rows the cropped output needs go through the horizontal pass (Pillow does the same for its box)
    const int y0 = cv.bounds[2 * crop_top];
    const int ylast = crop_top + out_h - 1;
    const int y1 = cv.bounds[2 * ylast] + cv.bounds[2 * ylast + 1];
    const int ny = y1 - y0;
    uint8_t* tmp = (uint8_t*)(ws + off[4]);
    hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for((long long)ny * resize_w, 256)), dim3(256), 0, s, (const uint8_t*)rgb_hwc,
                       row_stride, y0, ny, resize_w, (const int*)(ws + off[0]), (const int*)(ws + off[1]), ch.ksize, tmp);
    int rc = seedmi_check_launch("resize_h");
    if (rc != SEEDMI_OK) return rc;
    const long long npx = (long long)out_h * out_w;
    if (out_is_fp32)
        hipLaunchKernelGGL(resize_v_norm_kernel<true>, dim3(grid_for(npx, 256)), dim3(256), 0, s, tmp, resize_w, y0,
                           (const int*)(ws + off[2]), (const int*)(ws + off[3]), cv.ksize, crop_top, crop_left, out_h, out_w,
                           mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out_chw, (uint8_t*)out_u8_hwc);
    else
        hipLaunchKernelGGL(resize_v_norm_kernel<false>, dim3(grid_for(npx, 256)), dim3(256), 0, s, tmp, resize_w, y0,
                           (const int*)(ws + off[2]), (const int*)(ws + off[3]), cv.ksize, crop_top, crop_left, out_h, out_w,
                           mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out_chw, (uint8_t*)out_u8_hwc);
    return seedmi_check_launch("resize_v_norm");
}
