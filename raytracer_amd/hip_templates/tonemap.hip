// Tone mapping of the radiance accumulator on the device (reference imgutils.py:60-80, the
// flare-free path): 8-bit RGB out of the float sums without bringing 16 B per pixel to the host.
// Arithmetic is binary64 like the numpy pipeline it replaces.  `img` is in ray order; `order`
// gives (row, col) of every ray; `base` (optional) = earlier sums to add (--append).
typedef unsigned int uint;

static __device__ __forceinline__ double clr_tm_clean(double v) {
    // numpy.nan_to_num: NaN -> 0, +-inf -> +-largest finite
    if (v != v) return 0.0;
    if (v > 1.7976931348623157e308) return 1.7976931348623157e308;
    if (v < -1.7976931348623157e308) return -1.7976931348623157e308;
    return v;
}

static __device__ __forceinline__ void clr_tm_pixel(const float4 *img, const double *base, int k, double rgb[3]) {
    const float4 v = img[k];
    rgb[0] = v.x; rgb[1] = v.y; rgb[2] = v.z;
    if (base) { rgb[0] += base[3*k]; rgb[1] += base[3*k + 1]; rgb[2] += base[3*k + 2]; }
    for (int c = 0; c < 3; ++c) rgb[c] = clr_tm_clean(rgb[c]);
}

// per block: minimum and sum of the lightness (r + g + b) / 3
extern "C" __global__ void __launch_bounds__(256)
clr_tm_reduce(const float4 * __restrict__ img, const double * __restrict__ base, const int n_pixels,
              double * __restrict__ block_min, double * __restrict__ block_sum)
{
    __shared__ double s_min[256], s_sum[256];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    double lo = 1.7976931348623157e308, sum = 0.0;
    if (k < n_pixels) {
        double rgb[3];
        clr_tm_pixel(img, base, k, rgb);
        lo = sum = (rgb[0] + rgb[1] + rgb[2]) / 3.0;
    }
    s_min[threadIdx.x] = lo;
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int step = 128; step > 0; step >>= 1) {
        if ((int)threadIdx.x < step) {
            s_min[threadIdx.x] = fmin(s_min[threadIdx.x], s_min[threadIdx.x + step]);
            s_sum[threadIdx.x] += s_sum[threadIdx.x + step];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { block_min[blockIdx.x] = s_min[0]; block_sum[blockIdx.x] = s_sum[0]; }
}

// out[(row * width + col) * 3 + c] = 8-bit value
extern "C" __global__ void __launch_bounds__(256)
clr_tm_encode(const float4 * __restrict__ img, const double * __restrict__ base,
              const int2 * __restrict__ order, const int n_pixels, const int width,
              const double floor_value, const double scale, const double inv_gamma,
              unsigned char * __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pixels) return;
    double rgb[3];
    clr_tm_pixel(img, base, k, rgb);
    const int2 rc = order[k];
    unsigned char *dst = out + ((size_t)rc.x * width + rc.y) * 3;
    for (int c = 0; c < 3; ++c) {
        double v = (rgb[c] - floor_value) * scale;
        v = v > 0.0 ? v : 0.0;
        v = pow(v, inv_gamma);
        v = v < 1.0 ? v : 1.0;
        dst[c] = (unsigned char)(v * 255.0);
    }
}
